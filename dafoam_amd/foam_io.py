"""OpenFOAM ASCII case IO for the hot path's inputs: constant/polyMesh/{points,faces,owner,
neighbour,boundary}, 0/<field> files (internalField + boundaryField) and constant/transportProperties:nu.

The reference gets these through OpenFOAM's own readers when DASolver constructs argList/Time/fvMesh
(reference src/include/createMeshPython.H / createFields*.H via DASolver::initSolver, e.g. DASimpleFoam.C:81-121);
this module makes the same data available as a :class:`dafoam_amd.meshgen.FoamCase` so that an existing case
directory can drive the GPU path.  ASCII only (uncompressed), uniform or nonuniform List fields, the patch-field types
of SURVEY.md Appendix B (fixedValue, zeroGradient, inletOutlet, symmetry/symmetryPlane, noSlip, calculated,
nutUSpaldingWallFunction, nutLowReWallFunction, fixedValue nut = 0).
"""
from __future__ import annotations

import os
import re

import numpy as np

from .meshgen import (
    BC_FIXED_VALUE, BC_INLET_OUTLET, BC_SYMMETRY, BC_ZERO_GRADIENT, NUT_CALCULATED, NUT_LOWRE_WALL, NUT_SPALDING_WALL, NUT_SYMMETRY,
    FoamCase, Patch, PolyMesh,
)

_HEADER = """/*--------------------------------*- C++ -*----------------------------------*\\
| dafoam_amd foam_io                                                          |
\\*---------------------------------------------------------------------------*/
FoamFile
{{
    version     2.0;
    format      ascii;
    class       {cls};
    location    "{loc}";
    object      {obj};
}}
// * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * * //

"""


def _strip(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*?$", "", text, flags=re.M)
    m = re.search(r"FoamFile\s*\{.*?\}", text, flags=re.S)
    if m:
        text = text[: m.start()] + text[m.end():]
    return text


def _list_body(text):
    """'N ( ... )' -> (N, body string)"""
    m = re.search(r"(\d+)\s*\(", text)
    if not m:
        raise ValueError("no list found")
    n = int(m.group(1))
    start = m.end()
    depth, i = 1, start
    while depth and i < len(text):
        c = text[i]
        depth += c == "("
        depth -= c == ")"
        i += 1
    return n, text[start : i - 1]


def read_points(path):
    n, body = _list_body(_strip(open(path).read()))
    a = np.array(re.sub(r"[()]", " ", body).split(), dtype=np.float64).reshape(-1, 3)
    assert a.shape[0] == n, f"{path}: expected {n} points, found {a.shape[0]}"
    return a


def write_points(path, points):
    """a polyMesh `points` file (vectorField) at `path` (directories created)"""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    with open(path, "w") as f:
        f.write(_HEADER.format(cls="vectorField", loc=os.path.dirname(path), obj="points"))
        f.write(f"{pts.shape[0]}\n(\n" + "\n".join("(%.17g %.17g %.17g)" % tuple(p) for p in pts) + "\n)\n")


def read_labels(path):
    n, body = _list_body(_strip(open(path).read()))
    a = np.array(body.split(), dtype=np.int64)
    assert a.size == n, f"{path}: expected {n} labels, found {a.size}"
    return a.astype(np.int32)


def read_faces(path):
    n, body = _list_body(_strip(open(path).read()))
    ptr, pts = [0], []
    for m in re.finditer(r"(\d+)\s*\(([^()]*)\)", body):
        k = int(m.group(1))
        v = m.group(2).split()
        assert len(v) == k
        pts.extend(int(x) for x in v)
        ptr.append(len(pts))
    assert len(ptr) - 1 == n, f"{path}: expected {n} faces, found {len(ptr)-1}"
    return np.array(ptr, dtype=np.int32), np.array(pts, dtype=np.int32)


def _parse_dict_entries(body):
    """name { key value; ... } blocks -> dict of dicts (values as strings)."""
    out = {}
    for m in re.finditer(r"(\"[^\"]+\"|[A-Za-z_][\w.:\-]*)\s*\{([^{}]*)\}", body):
        d = {}
        for e in re.finditer(r"([A-Za-z_]\w*)\s+([^;]*);", m.group(2)):
            d[e.group(1)] = e.group(2).strip()
        out[m.group(1).strip('"')] = d
    return out


def read_boundary(path):
    n, body = _list_body(_strip(open(path).read()))
    ent = _parse_dict_entries(body)
    assert len(ent) == n, f"{path}: expected {n} patches, found {len(ent)}"
    patches = []
    for name, d in ent.items():
        t = d.get("type", "patch")
        t = {"symmetryPlane": "symmetry", "empty": "symmetry"}.get(t, t)
        if t not in ("patch", "wall", "symmetry", "cyclic"):
            raise NotImplementedError(f"patch type {t} of {name} is outside the hot path (processor / AMI patches: see DESIGN.md)")
        pt = Patch(name, t, int(d["startFace"]), int(d["nFaces"]))
        if t == "cyclic":
            pt.neighbour = d["neighbourPatch"].strip()
            pt._transform = {k: d[k].strip() for k in ("transform", "rotationAxis", "rotationCentre", "separationVector") if k in d}
        patches.append(pt)
    patches.sort(key=lambda p: p.start)
    return patches


def _cyclic_rotations(mesh: PolyMesh):
    """forwardT of rotational cyclic pairs (cyclicPolyPatch: rotationAxis / rotationCentre; the angle follows from the
    first face pair): the rotation that carries neighbour-side vectors into this side's frame."""
    by_name = {p.name: p for p in mesh.patches}
    for pt in mesh.patches:
        tr = getattr(pt, "_transform", None)
        if pt.type != "cyclic" or not tr or tr.get("transform", "translational") != "rotational":
            continue
        axis = np.array(re.sub(r"[()]", " ", tr["rotationAxis"]).split(), dtype=np.float64)
        axis /= np.linalg.norm(axis)
        centre = np.array(re.sub(r"[()]", " ", tr.get("rotationCentre", "(0 0 0)")).split(), dtype=np.float64)
        nb = by_name[pt.neighbour]

        def fc(f):
            v = mesh.face_pts[mesh.face_ptr[f] : mesh.face_ptr[f + 1]]
            return mesh.points[v].mean(0) - centre

        a, b = fc(pt.start), fc(nb.start)  # this side, neighbour side
        a, b = a - axis * (a @ axis), b - axis * (b @ axis)
        ang = np.arctan2(np.cross(b, a) @ axis, b @ a)  # rotating the neighbour side by `ang` about the axis lands on this side
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        pt.rotation = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def read_polymesh(case_dir):
    pm = os.path.join(case_dir, "constant", "polyMesh")
    fptr, fpts = read_faces(os.path.join(pm, "faces"))
    mesh = PolyMesh(points=read_points(os.path.join(pm, "points")), face_ptr=fptr, face_pts=fpts, owner=read_labels(os.path.join(pm, "owner")),
                    neighbour=read_labels(os.path.join(pm, "neighbour")), patches=read_boundary(os.path.join(pm, "boundary")))
    nB = sum(p.size for p in mesh.patches)
    assert mesh.n_internal_faces + nB == mesh.n_faces, "boundary does not cover all boundary faces"
    _cyclic_rotations(mesh)
    return mesh


def write_polymesh(case_dir, mesh: PolyMesh):
    pm = os.path.join(case_dir, "constant", "polyMesh")
    os.makedirs(pm, exist_ok=True)
    with open(os.path.join(pm, "points"), "w") as f:
        f.write(_HEADER.format(cls="vectorField", loc="constant/polyMesh", obj="points"))
        f.write(f"{mesh.n_points}\n(\n" + "\n".join("(%.17g %.17g %.17g)" % tuple(p) for p in mesh.points) + "\n)\n")
    with open(os.path.join(pm, "faces"), "w") as f:
        f.write(_HEADER.format(cls="faceList", loc="constant/polyMesh", obj="faces"))
        rows = []
        for k in range(mesh.n_faces):
            v = mesh.face_pts[mesh.face_ptr[k] : mesh.face_ptr[k + 1]]
            rows.append(f"{len(v)}(" + " ".join(str(int(x)) for x in v) + ")")
        f.write(f"{mesh.n_faces}\n(\n" + "\n".join(rows) + "\n)\n")
    for nm, arr in (("owner", mesh.owner), ("neighbour", mesh.neighbour)):
        with open(os.path.join(pm, nm), "w") as f:
            f.write(_HEADER.format(cls="labelList", loc="constant/polyMesh", obj=nm))
            f.write(f"{arr.size}\n(\n" + "\n".join(str(int(x)) for x in arr) + "\n)\n")
    with open(os.path.join(pm, "boundary"), "w") as f:
        f.write(_HEADER.format(cls="polyBoundaryMesh", loc="constant/polyMesh", obj="boundary"))
        f.write(f"{len(mesh.patches)}\n(\n")
        by_name = {p.name: p for p in mesh.patches}
        for p in mesh.patches:
            extra = ""
            if p.type == "cyclic":
                extra = f"        neighbourPatch  {p.neighbour};\n"
                if p.rotation is not None:  # axis-angle of forwardT
                    Q = np.asarray(p.rotation, dtype=np.float64).reshape(3, 3)
                    ax = np.array([Q[2, 1] - Q[1, 2], Q[0, 2] - Q[2, 0], Q[1, 0] - Q[0, 1]])
                    ax = ax / np.linalg.norm(ax) if np.linalg.norm(ax) > 0 else np.array([1.0, 0.0, 0.0])
                    if ax[np.argmax(np.abs(ax) > 1e-12)] < 0:
                        ax = -ax  # one sign for both patches of the pair (the reader takes the angle's sign from the face pair)

                    def fcr(f):
                        return mesh.points[mesh.face_pts[mesh.face_ptr[f] : mesh.face_ptr[f + 1]]].mean(0)

                    # a point on the axis: a - c = Q (b - c) for the first face pair (a this side, b the neighbour side);
                    # I - Q is singular along the axis, the least-squares solution is the axis point nearest the origin
                    a, b = fcr(p.start), fcr(by_name[p.neighbour].start)
                    cen = np.linalg.lstsq(np.eye(3) - Q, a - Q @ b, rcond=None)[0]
                    cen[np.abs(cen) < 1e-14 * max(1.0, np.abs(mesh.points).max())] = 0.0
                    extra += ("        transform       rotational;\n        rotationAxis    (%.17g %.17g %.17g);\n        rotationCentre  (%.17g %.17g %.17g);\n"
                              % (tuple(ax) + tuple(cen)))
                else:
                    def fc(f):
                        return mesh.points[mesh.face_pts[mesh.face_ptr[f] : mesh.face_ptr[f + 1]]].mean(0)
                    sep = fc(by_name[p.neighbour].start) - fc(p.start)
                    extra += "        transform       translational;\n        separationVector (%.17g %.17g %.17g);\n" % tuple(sep)
            f.write(f"    {p.name}\n    {{\n        type            {p.type};\n{extra}        nFaces          {p.size};\n        startFace       {p.start};\n    }}\n")
        f.write(")\n")


# ----------------------------------------------------------------------------- fields
class _NonUniform:
    """Marker for a patch `value` that is not a uniform constant (nonuniform List, $internalField, ...): accepted where the
    patch type does not consume it (zeroGradient, calculated, wall functions, ...), rejected where it would be needed."""

    def __init__(self, text):
        self.text = text[:40]


def _parse_value(s, ncomp):
    s = s.strip()
    if s.startswith("uniform"):
        v = np.array(re.sub(r"[()]", " ", s[len("uniform"):]).split(), dtype=np.float64)
        return v if ncomp == 3 else float(v[0])
    return _NonUniform(s)


def read_field(path, n_cells, ncomp):
    """returns (internal (n_cells[,3]) array, {patch: dict(type=..., value=...)})"""
    text = _strip(open(path).read())
    m = re.search(r"internalField\s+(uniform\s+[^;]+|nonuniform\s+List<\w+>\s*)", text)
    if not m:
        raise ValueError(f"{path}: no internalField")
    if m.group(1).startswith("uniform"):
        v = _parse_value(m.group(1), ncomp)
        internal = np.tile(np.atleast_1d(v), (n_cells, 1)) if ncomp == 3 else np.full(n_cells, v)
    else:
        n, body = _list_body(text[m.end():])
        internal = np.array(re.sub(r"[()]", " ", body).split(), dtype=np.float64)
        internal = internal.reshape(n, 3) if ncomp == 3 else internal
        assert internal.shape[0] == n_cells
    bm = re.search(r"boundaryField\s*\{", text)
    depth, i = 1, bm.end()
    while depth:
        depth += text[i] == "{"
        depth -= text[i] == "}"
        i += 1
    bfield = {}
    for name, d in _parse_dict_entries(text[bm.end() : i - 1]).items():
        e = {"type": d.get("type", "zeroGradient")}
        for key in ("value", "inletValue"):
            if key in d:
                e[key] = _parse_value(d[key], ncomp)
        bfield[name] = e
    return internal, bfield


_SCALAR_BC = {"fixedValue": BC_FIXED_VALUE, "zeroGradient": BC_ZERO_GRADIENT, "inletOutlet": BC_INLET_OUTLET, "symmetry": BC_SYMMETRY,
              "symmetryPlane": BC_SYMMETRY, "empty": BC_SYMMETRY, "calculated": BC_ZERO_GRADIENT}
_NUT_BC = {"calculated": NUT_CALCULATED, "nutUSpaldingWallFunction": NUT_SPALDING_WALL, "nutUSpaldingWallFunctionDF": NUT_SPALDING_WALL,
           "nutLowReWallFunction": NUT_LOWRE_WALL, "symmetry": NUT_SYMMETRY, "symmetryPlane": NUT_SYMMETRY, "empty": NUT_SYMMETRY,
           "zeroGradient": NUT_CALCULATED}


def _bc_entry(e, vec):
    t = e["type"]
    if t == "noSlip":
        return (BC_FIXED_VALUE, (0.0, 0.0, 0.0) if vec else 0.0)
    if t not in _SCALAR_BC:
        raise NotImplementedError(f"patch field type {t} is outside the hot path")
    code = _SCALAR_BC[t]
    v = e.get("inletValue" if t == "inletOutlet" else "value", np.zeros(3) if vec else 0.0)
    if isinstance(v, _NonUniform):
        if code in (BC_FIXED_VALUE, BC_INLET_OUTLET):  # the value IS the boundary condition: must be a constant here
            raise NotImplementedError(f"patch field type {t} with a non-uniform value is outside the hot path: {v.text}")
        v = np.zeros(3) if vec else 0.0  # zeroGradient / calculated / symmetry restart files carry a value nobody reads
    return (code, tuple(np.atleast_1d(v)) if vec else float(v))


# ---- system/fvSchemes, system/fvSolution ------------------------------------------------------------------------------------
# The reference takes its discretisation from the case (DAResidualSimpleFoam.C:123-132 builds fvm::div(phi, U) etc. through the
# run-time selected schemes, DASpalartAllmaras.C:428-447 likewise; relaxation through fvSolution).  The HIP kernels implement ONE
# scheme set (DESIGN.md section 3 - the set of the reference's own regression cases); the reader below makes that explicit: the
# relaxation factors and the SIMPLE switches of the case are honoured, every scheme entry is compared with the implemented set
# and a case that asks for anything else is rejected with the list of offending entries instead of being run with other numerics.
IMPLEMENTED_SCHEMES = {
    "ddtSchemes": {"default": ["steadyState"]},
    "gradSchemes": {"default": ["Gauss linear"], "grad(U)": ["Gauss linear"], "grad(p)": ["Gauss linear"], "grad(nuTilda)": ["Gauss linear"]},
    "divSchemes": {
        "default": ["none"],
        "div(phi,U)": ["bounded Gauss linearUpwindV grad(U)"],
        "div(phi,nuTilda)": ["bounded Gauss upwind"],
        "div(pc)": ["bounded Gauss upwind"],
        "div((nuEff*dev2(T(grad(U)))))": ["Gauss linear"],
        # compressible solvers
        "div(phi,T)": ["bounded Gauss upwind"], "div(phi,h)": ["bounded Gauss upwind"], "div(phi,e)": ["bounded Gauss upwind"],
        "div(phi,K)": ["bounded Gauss upwind"], "div(phi,Ekp)": ["bounded Gauss upwind"], "div(phid,p)": ["Gauss upwind", "bounded Gauss upwind"],
        "div(((rho*nuEff)*dev2(T(grad(U)))))": ["Gauss linear"],
    },
    "laplacianSchemes": {"default": ["Gauss linear corrected"]},
    "interpolationSchemes": {"default": ["linear"]},
    "snGradSchemes": {"default": ["corrected"]},
}


def _dict_block(text, name):
    """body of `name { ... }` at any depth of an (already comment-stripped) OpenFOAM dictionary, or None"""
    m = re.search(r"(?<![\w.:-])" + re.escape(name) + r"\s*\{", text)
    if not m:
        return None
    depth, i = 1, m.end()
    while depth and i < len(text):
        depth += text[i] == "{"
        depth -= text[i] == "}"
        i += 1
    return text[m.end() : i - 1]


def _flat_entries(body):
    """`key value...;` entries of a dictionary body (sub-dictionaries skipped); keys may be quoted regular expressions"""
    out, i = {}, 0
    body = re.sub(r"\{[^{}]*\}", "", body)  # (one level of nesting is all these dictionaries have)
    for m in re.finditer(r'("[^"]+"|[^\s;{}]+)\s+([^;{}]*);', body):
        out[m.group(1).strip('"')] = " ".join(m.group(2).split())
    return out


def read_fv_schemes(case_dir):
    """system/fvSchemes -> {section: {entry: scheme string}} (None if the file does not exist)"""
    path = os.path.join(case_dir, "system", "fvSchemes")
    if not os.path.exists(path):
        return None
    text = _strip(open(path).read())
    out = {}
    for sec in ("ddtSchemes", "gradSchemes", "divSchemes", "laplacianSchemes", "interpolationSchemes", "snGradSchemes", "wallDist"):
        b = _dict_block(text, sec)
        if b is not None:
            out[sec] = _flat_entries(b)
    return out


def check_schemes(schemes):
    """Entries of a parsed fvSchemes that differ from the scheme set the kernels implement: list of "section/entry: value"."""
    bad = []
    for sec, allowed in IMPLEMENTED_SCHEMES.items():
        for key, val in (schemes.get(sec) or {}).items():
            ok = allowed.get(key, allowed.get("default") if key != "default" else None)
            if ok is None or val not in ok:
                bad.append(f"{sec}/{key}: {val}")
    wd = (schemes.get("wallDist") or {}).get("method")
    if wd is not None and wd not in ("meshWaveFrozen", "meshWave"):
        bad.append(f"wallDist/method: {wd}")
    return bad


def read_fv_solution(case_dir):
    """system/fvSolution -> dict(relax={field: factor}, consistent, transonic, nNonOrthogonalCorrectors) (None without the file).
    relaxationFactors/equations (and fields) entries may be quoted regular expressions like "(U|T|nuTilda)"."""
    path = os.path.join(case_dir, "system", "fvSolution")
    if not os.path.exists(path):
        return None
    text = _strip(open(path).read())
    out = {"relax": {}, "relax_fields": {}, "consistent": False, "transonic": False, "nNonOrthogonalCorrectors": 0}
    simple = _dict_block(text, "SIMPLE")
    if simple is not None:
        e = _flat_entries(simple)
        truth = lambda v: str(v).lower() in ("true", "yes", "on", "1")  # noqa: E731
        out["consistent"] = truth(e.get("consistent", "false"))
        out["transonic"] = truth(e.get("transonic", "false"))
        out["nNonOrthogonalCorrectors"] = int(e.get("nNonOrthogonalCorrectors", 0))
    rf = _dict_block(text, "relaxationFactors")
    if rf is not None:
        for sub, dst in (("equations", "relax"), ("fields", "relax_fields")):
            b = _dict_block(rf, sub)
            if b is None:
                continue
            # OpenFOAM's dictionary lookup: an exact keyword wins; among the regular-expression keys the LAST matching one
            ent = _flat_entries(b)
            for name in ("U", "nuTilda", "T", "h", "e", "p", "rho"):
                if name in ent:
                    out[dst][name] = float(ent[name])
                    continue
                for key, val in ent.items():
                    try:
                        hit = re.fullmatch(key, name) is not None
                    except re.error:  # an unquoted keyword that is not a valid pattern can only match literally
                        hit = False
                    if hit:
                        out[dst][name] = float(val)
    return out


def apply_system_dicts(case: FoamCase, case_dir, strict=True):
    """Honour system/fvSolution (equation relaxation factors, SIMPLE consistent / transonic) and verify system/fvSchemes against
    the implemented scheme set.  strict: a case that asks for other schemes raises NotImplementedError naming the entries.  Field
    relaxation factors (p) and nNonOrthogonalCorrectors are parameters of the SIMPLE ITERATION, not of its fixed point R(W) = 0
    that this library evaluates: they are read (read_fv_solution) and deliberately not used."""
    sch = read_fv_schemes(case_dir)
    if sch is not None:
        bad = check_schemes(sch)
        if bad and strict:
            raise NotImplementedError("system/fvSchemes asks for schemes the GPU kernels do not implement (DESIGN.md section 3): " + "; ".join(bad))
        case.scheme_mismatches = bad
    sol = read_fv_solution(case_dir)
    if sol is not None:
        relax = dict(case.relax)
        for k in ("U", "nuTilda"):
            if k in sol["relax"]:
                relax[k] = sol["relax"][k]
        for k in ("T", "h", "e"):
            if k in sol["relax"]:
                relax["T"] = sol["relax"][k]
        case.relax = relax
        case.simple_consistent = bool(sol["consistent"])
        case.transonic = bool(sol["transonic"])
    return case


def write_system_dicts(case_dir, case: FoamCase):
    """system/fvSchemes + system/fvSolution of the scheme set the kernels implement, with the case's relaxation factors."""
    os.makedirs(os.path.join(case_dir, "system"), exist_ok=True)
    comp = case.solver_name != "DASimpleFoam"
    with open(os.path.join(case_dir, "system", "fvSchemes"), "w") as f:
        f.write(_HEADER.format(cls="dictionary", loc="system", obj="fvSchemes"))
        f.write("ddtSchemes { default steadyState; }\ngradSchemes { default Gauss linear; }\ndivSchemes\n{\n    default none;\n")
        f.write("    div(phi,U) bounded Gauss linearUpwindV grad(U);\n    div(phi,nuTilda) bounded Gauss upwind;\n    div(pc) bounded Gauss upwind;\n")
        if comp:
            f.write("    div(phi,T) bounded Gauss upwind;\n    div(phi,h) bounded Gauss upwind;\n    div(phi,K) bounded Gauss upwind;\n"
                    "    div(phid,p) Gauss upwind;\n    div(((rho*nuEff)*dev2(T(grad(U))))) Gauss linear;\n")
        else:
            f.write("    div((nuEff*dev2(T(grad(U))))) Gauss linear;\n")
        f.write("}\nlaplacianSchemes { default Gauss linear corrected; }\ninterpolationSchemes { default linear; }\nsnGradSchemes { default corrected; }\n"
                "wallDist { method meshWaveFrozen; }\n")
    with open(os.path.join(case_dir, "system", "fvSolution"), "w") as f:
        f.write(_HEADER.format(cls="dictionary", loc="system", obj="fvSolution"))
        f.write("SIMPLE\n{\n    nNonOrthogonalCorrectors 0;\n")
        f.write(f"    consistent {'true' if getattr(case, 'simple_consistent', False) else 'false'};\n")
        f.write(f"    transonic {'true' if getattr(case, 'transonic', False) else 'false'};\n}}\n")
        f.write("relaxationFactors\n{\n    equations\n    {\n")
        f.write(f"        U {case.relax.get('U', 0.7):.17g};\n        nuTilda {case.relax.get('nuTilda', 0.7):.17g};\n")
        if comp or getattr(case, "has_T", False):
            f.write(f"        \"(T|h|e)\" {case.relax.get('T', 1.0):.17g};\n")
        f.write("    }\n}\n")


def read_case(case_dir, solver_name="DASimpleFoam", time="0", y_wall=None, strict_schemes=True) -> FoamCase:
    """DASimpleFoam / DARhoSimpleFoam case directory -> FoamCase (phi = interp(U).Sf if 0/phi is absent).  system/fvSolution
    (relaxation factors, SIMPLE switches) is honoured and system/fvSchemes is verified against the implemented scheme set when
    the files exist (apply_system_dicts)."""
    from .meshgen import _InputGeometry, wall_distance_exact as wall_distance

    mesh = read_polymesh(case_dir)
    N, F, nIF = mesh.n_cells, mesh.n_faces, mesh.n_internal_faces
    t = os.path.join(case_dir, time)
    U, bU = read_field(os.path.join(t, "U"), N, 3)
    p, bp = read_field(os.path.join(t, "p"), N, 1)
    nuT, bn = read_field(os.path.join(t, "nuTilda"), N, 1)
    _, bnut = read_field(os.path.join(t, "nut"), N, 1)
    bcs = {}
    for pt in mesh.patches:
        nm = pt.name
        e = {"U": _bc_entry(bU[nm], True), "p": _bc_entry(bp[nm], False), "nuTilda": _bc_entry(bn[nm], False)}
        nt = bnut[nm]["type"]
        if nt == "fixedValue":
            e["nut"] = (NUT_LOWRE_WALL, 0.0)
        elif nt in _NUT_BC:
            e["nut"] = (_NUT_BC[nt], 0.0)
        else:
            raise NotImplementedError(f"nut patch type {nt}")
        bcs[nm] = e
    tp = _strip(open(os.path.join(case_dir, "constant", "transportProperties")).read())
    m = re.search(r"\bnu\s+(?:\[[^\]]*\]\s*)?([-+0-9.eE]+)\s*;", tp)
    nu = float(m.group(1)) if m else 1.5e-5
    g = _InputGeometry(mesh)
    if y_wall is None:
        y_wall = wall_distance(mesh, g.C, g.Cf, g.Sf)
    case = FoamCase(mesh=mesh, solver_name=solver_name, nu=nu, bcs=bcs, y_wall=y_wall)
    # phi: the face-flux state, internal AND boundary faces (DAIndex.C:109-112).  Values the file does not give (no phi
    # file, a patch without a value) are derived from the patch velocity like OpenFOAM's createPhi does.
    own, nei = mesh.owner, mesh.neighbour
    Uf = g.w[:, None] * U[own[:nIF]] + (1 - g.w[:, None]) * U[nei]
    phi = np.zeros(F)
    phi[:nIF] = np.einsum("ij,ij->i", Uf, g.Sf[:nIF])
    for pt in mesh.patches:
        sl = slice(pt.start, pt.start + pt.size)
        code, val = bcs[pt.name]["U"]
        if code == BC_FIXED_VALUE:
            phi[sl] = g.Sf[sl] @ np.asarray(val, dtype=float)
        elif code != BC_SYMMETRY:
            phi[sl] = np.einsum("ij,ij->i", U[own[sl]], g.Sf[sl])
    phi_path = os.path.join(t, "phi")
    if os.path.exists(phi_path):
        text = _strip(open(phi_path).read())
        m = re.search(r"internalField\s+(uniform\s+[^;]+|nonuniform\s+List<scalar>\s*)", text)
        if m.group(1).startswith("uniform"):
            phi[:nIF] = float(m.group(1).split()[1])
        else:
            n, body = _list_body(text[m.end():])
            phi[:nIF] = np.array(body.split(), dtype=np.float64)
        bm = re.search(r"boundaryField\s*\{", text)
        if bm:
            for pt in mesh.patches:
                if not pt.size:
                    continue
                pm = re.search(r"(?<![\w.:-])" + re.escape(pt.name) + r"\s*\{", text[bm.end():])
                if not pm:
                    continue
                blk = text[bm.end() + pm.end():]
                blk = blk[: blk.index("}")]
                vm = re.search(r"\bvalue\s+(uniform\s+[-+0-9.eE]+|nonuniform\s+List<scalar>\s*)", blk)
                if not vm:
                    continue
                sl = slice(pt.start, pt.start + pt.size)
                if vm.group(1).startswith("uniform"):
                    phi[sl] = float(vm.group(1).split()[1])
                else:
                    cnt, body = _list_body(blk[vm.end():])
                    if cnt == pt.size:
                        phi[sl] = np.array(body.split(), dtype=np.float64)
    case.states = np.concatenate([U.ravel(), p, nuT, phi])
    return apply_system_dicts(case, case_dir, strict=strict_schemes)


def write_case(case_dir, case: FoamCase, time="0"):
    """Write a DASimpleFoam FoamCase as an OpenFOAM ASCII case (mesh, 0/U p nuTilda nut, transportProperties)."""
    mesh = case.mesh
    write_polymesh(case_dir, mesh)
    N = mesh.n_cells
    W = case.states
    U, p, nuT = W[: 3 * N].reshape(N, 3), W[3 * N : 4 * N], W[4 * N : 5 * N]
    tdir = os.path.join(case_dir, time)
    os.makedirs(tdir, exist_ok=True)
    names = {BC_FIXED_VALUE: "fixedValue", BC_ZERO_GRADIENT: "zeroGradient", BC_INLET_OUTLET: "inletOutlet", BC_SYMMETRY: "symmetry"}
    nutn = {NUT_CALCULATED: "calculated", NUT_LOWRE_WALL: "nutLowReWallFunction", NUT_SPALDING_WALL: "nutUSpaldingWallFunction", NUT_SYMMETRY: "symmetry"}

    def fmt(v):
        return "(%.17g %.17g %.17g)" % tuple(v) if np.ndim(v) else "%.17g" % v

    def write(name, cls, internal, field):
        with open(os.path.join(tdir, name), "w") as f:
            f.write(_HEADER.format(cls=cls, loc=time, obj=name))
            f.write("dimensions      [0 0 0 0 0 0 0];\n\n")
            if internal is None:
                f.write("internalField   uniform 0;\n\n")
            else:
                typ = "vector" if internal.ndim == 2 else "scalar"
                f.write(f"internalField   nonuniform List<{typ}>\n{internal.shape[0]}\n(\n" + "\n".join(fmt(v) for v in internal) + "\n)\n;\n\n")
            f.write("boundaryField\n{\n")
            for pt in mesh.patches:
                code, val = case.bcs[pt.name][field]
                f.write(f"    {pt.name}\n    {{\n")
                if field == "nut":
                    f.write(f"        type            {nutn[code]};\n")
                    if code != NUT_SYMMETRY:
                        f.write("        value           uniform 0;\n")
                else:
                    f.write(f"        type            {names[code]};\n")
                    if code == BC_FIXED_VALUE:
                        f.write(f"        value           uniform {fmt(np.asarray(val)) if np.ndim(val) else fmt(val)};\n")
                    if code == BC_INLET_OUTLET:
                        f.write(f"        inletValue      uniform {fmt(np.asarray(val)) if np.ndim(val) else fmt(val)};\n")
                        f.write(f"        value           uniform {fmt(np.asarray(val)) if np.ndim(val) else fmt(val)};\n")
                f.write("    }\n")
            f.write("}\n")

    write("U", "volVectorField", U, "U")
    write("p", "volScalarField", p, "p")
    write("nuTilda", "volScalarField", nuT, "nuTilda")
    write("nut", "volScalarField", None, "nut")
    os.makedirs(os.path.join(case_dir, "constant"), exist_ok=True)
    with open(os.path.join(case_dir, "constant", "transportProperties"), "w") as f:
        f.write(_HEADER.format(cls="dictionary", loc="constant", obj="transportProperties"))
        f.write(f"transportModel  Newtonian;\n\nnu              [0 2 -1 0 0 0 0] {case.nu:.17g};\n")
    write_system_dicts(case_dir, case)
    nIF = mesh.n_internal_faces
    with open(os.path.join(tdir, "phi"), "w") as f:
        f.write(_HEADER.format(cls="surfaceScalarField", loc=time, obj="phi"))
        f.write(f"dimensions      [0 3 -1 0 0 0 0];\n\ninternalField   nonuniform List<scalar>\n{nIF}\n(\n" + "\n".join("%.17g" % v for v in W[5 * N : 5 * N + nIF]) + "\n)\n;\n\n")
        f.write("boundaryField\n{\n")
        for pt in mesh.patches:
            vals = W[5 * N + pt.start : 5 * N + pt.start + pt.size]
            f.write(f"    {pt.name}\n    {{\n        type            calculated;\n        value           nonuniform List<scalar>\n{pt.size}\n(\n"
                    + "\n".join("%.17g" % v for v in vals) + "\n)\n;\n    }\n")
        f.write("}\n")


def write_adjoint_fields(case_dir, case: FoamCase, function, write_time, psi, state_blocks):
    """DASolver::writeAdjointFields (reference DASolver.C:4055-4160): the adjoint vector as OpenFOAM fields
    `adjoint_<function>_<state>` in <case_dir>/<write_time>/ (cell states: the state's patch types with zero-gradient
    values, i.e. calculated from the cell values; phi: surfaceScalarField with its boundary values).
    state_blocks: [(name, "vec"|"scl"|"face", offset, size)] in "state" ordering."""
    mesh = case.mesh
    N, nIF = mesh.n_cells, mesh.n_internal_faces
    time = ("%g" % write_time) if not isinstance(write_time, str) else write_time
    tdir = os.path.join(case_dir, time)
    os.makedirs(tdir, exist_ok=True)
    written = []
    for name, kind, off, size in state_blocks:
        var = f"adjoint_{function}_{name}"
        blk = np.asarray(psi[off : off + size], dtype=np.float64)
        with open(os.path.join(tdir, var), "w") as f:
            if kind == "face":
                f.write(_HEADER.format(cls="surfaceScalarField", loc=time, obj=var))
                f.write("dimensions      [0 0 0 0 0 0 0];\n\n")
                f.write(f"internalField   nonuniform List<scalar>\n{nIF}\n(\n" + "\n".join("%.17g" % v for v in blk[:nIF]) + "\n)\n;\n\n")
                f.write("boundaryField\n{\n")
                for pt in mesh.patches:
                    vals = blk[pt.start : pt.start + pt.size]
                    f.write(f"    {pt.name}\n    {{\n        type            calculated;\n        value           nonuniform List<scalar>\n{pt.size}\n(\n"
                            + "\n".join("%.17g" % v for v in vals) + "\n)\n;\n    }\n")
                f.write("}\n")
            else:
                vec = kind == "vec"
                f.write(_HEADER.format(cls="volVectorField" if vec else "volScalarField", loc=time, obj=var))
                f.write("dimensions      [0 0 0 0 0 0 0];\n\n")
                if vec:
                    rows = "\n".join("(%.17g %.17g %.17g)" % tuple(v) for v in blk.reshape(N, 3))
                    f.write(f"internalField   nonuniform List<vector>\n{N}\n(\n{rows}\n)\n;\n\n")
                else:
                    f.write(f"internalField   nonuniform List<scalar>\n{N}\n(\n" + "\n".join("%.17g" % v for v in blk) + "\n)\n;\n\n")
                f.write("boundaryField\n{\n")
                for pt in mesh.patches:
                    ptype = "symmetry" if pt.type == "symmetry" else "zeroGradient"
                    f.write(f"    {pt.name}\n    {{\n        type            {ptype};\n    }}\n")
                f.write("}\n")
        written.append(os.path.join(tdir, var))
    return written
