#!/usr/bin/env python
"""Parity check against a REAL DAFoam run (test infrastructure; the one route to pinning parity with the reference).

    python tests/parity_from_dafoam_dump.py --case <OpenFOAM case dir> [--dump <dir>] [--time 0] [--function CD]
                                            [--engine gpu|oracle] [--tol 1e-6] [--solver DASimpleFoam]

Inputs, all produced by DAFoam itself on the SAME case (single rank, adjStateOrdering "state"):
  <case>/constant/polyMesh, <case>/<time>/{U,p,nuTilda,nut,phi}      the converged primal the adjoint was linearised about
  <dump>/dRdWT.bin, <dump>/dRdWTPC.bin      writeJacobians ["dRdWT"] (DASolver.C:1080-1085; PETSc binary, DAUtility.C:282-441)
  <dump>/dRdWColoring_1.bin                 the colouring cache (DAJacCon.C:1886-1975), optional
  <case>/<adjTime>/adjoint_<function>_<state>   writeAdjointFields (pyDAFoam.py:907-915, DASolver.C:4055-4160), optional
  <dump>/norms.json                         optional known answers, e.g. {"dRdWTv_0.001": 1732.238877108044} - the
                                            runUnitTests_DATurbModel.py:96,124 convention: ||dRdW^T (0.001*1)||_2

What is compared (relative differences, PASS if <= --tol):
  * dRdWT, dRdWTPC entry-wise per (state block x residual block): ||A_ref - A||_F / ||A_ref||_F, and the structural
    pattern (entries only one side has);
  * dRdW^T.v for v = 0.001*1 and for a seeded random v; the summed-norm known answer if given;
  * the reference colouring validated against this repo's connectivity (DAColoring::validateColoring rule);
  * psi: this repo's adjoint solve of (dRdWT_ref) psi = rhs recovered from the reference psi, per state block.

Engines: "gpu" = the product path (PYDAFOAM on an MI355X); "oracle" = the CPU restatement (oracle/), which lets the same
dump also pin the ORACLE.  `--self-test` writes this repo's own dumps for a synthetic channel and runs the comparison on
them (round trip through the on-disk formats; used by tests/test_host_cpu.py and the gpu tier).
"""
import argparse
import json
import os
import re
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from dafoam_amd import foam_io, petsc_io  # noqa: E402


def state_blocks(case):
    m = case.mesh
    N, F = m.n_cells, m.n_faces
    names = [("U", "vec", 3 * N), ("p", "scl", N)]
    if case.solver_name in ("DARhoSimpleFoam", "DATurboFoam") or getattr(case, "has_T", False):
        names.append(("T", "scl", N))
    names += [("nuTilda", "scl", N), ("phi", "face", F)]
    out, off = [], 0
    for nm, kind, size in names:
        out.append((nm, kind, off, size))
        off += size
    return out


def read_adjoint_fields(case_dir, time, function, case):
    """adjoint_<function>_<state> fields -> psi in "state" ordering (inverse of foam_io.write_adjoint_fields)."""
    m = case.mesh
    N, nIF = m.n_cells, m.n_internal_faces
    psi = np.zeros(sum(b[3] for b in state_blocks(case)))
    for nm, kind, off, size in state_blocks(case):
        path = os.path.join(case_dir, str(time), f"adjoint_{function}_{nm}")
        if kind != "face":
            internal, _ = foam_io.read_field(path, N, 3 if kind == "vec" else 1)
            psi[off : off + size] = np.asarray(internal).ravel()
            continue
        text = foam_io._strip(open(path).read())
        mm = re.search(r"internalField\s+nonuniform\s+List<scalar>\s*", text)
        n, body = foam_io._list_body(text[mm.end():])
        psi[off : off + nIF] = np.array(body.split(), dtype=np.float64)
        for pt in m.patches:
            pm = re.search(r"\b" + re.escape(pt.name) + r"\s*\{[^}]*?value\s+nonuniform\s+List<scalar>\s*", text, re.S)
            if pm and pt.size:
                _, pb = foam_io._list_body(text[pm.end():])
                psi[off + pt.start : off + pt.start + pt.size] = np.array(pb.split(), dtype=np.float64)
    return psi


class Engine:
    """R(W), dRdWT, dRdWTPC, connectivity and the adjoint solve of this repo, by the GPU product path or by the oracle."""

    def __init__(self, case, kind, norm_states):
        self.case, self.kind = case, kind
        if kind == "gpu":
            from dafoam_amd.pyDAFoam import PYDAFOAM
            from dafoam_amd.pyDASolvers import Mat

            self.D = PYDAFOAM(options={"solverName": case.solver_name, "normalizeStates": dict(norm_states),
                                       "adjEqnOption": {"gmresRelTol": 1e-10, "gmresMaxIters": 2000, "printInfo": 0},
                                       "jacLowerBounds": {"dRdW": 0.0, "dRdWPC": 0.0},
                                       # a DAFoam dump's dRdWTPC was assembled with the case's div(pc) = upwind: the reference PC semantics,
                                       # not this library's default blend (amd.pcUpwindBlend 0.5)
                                       "amd": {"pcUpwindBlend": 0.0}}, case=case)
            self.D.solver.runColoring()
            self._Mat = Mat
        else:
            from oracle import jacobian as J
            from oracle.foam_mesh import Geometry

            self.J, self.g = J, Geometry(case.mesh)
            self.sc = J.state_scales(case, self.g, norm_states)
            self.con = J.connectivity(case, self.g)
            self.col, _ = J.greedy_coloring(self.con)

    def matrix(self, isPC):
        if self.kind == "gpu":
            M = self._Mat()
            self.D.solver.calcdRdWT(isPC, M)
            A = M.to_scipy()
            M.destroy()
            return A.tocsr()
        con = self.J.connectivity(self.case, self.g, isPC=True) if isPC else self.con
        return self.J.jacobian_colored(self.case, self.g, self.case.states, con, self.col, self.sc, mode="fd" if isPC else "cs", isPC=bool(isPC),
                                       lower_bound=0).tocsr()

    def connectivity(self):
        return self.D.solver.getConnectivity(0) if self.kind == "gpu" else self.con

    def solve(self, rhs):
        if self.kind == "gpu":
            return self.D.solveAdjoint(rhs)
        import scipy.sparse.linalg as spla

        return spla.spsolve(self.matrix(0).tocsc(), rhs), 0


def block_report(A_ref, A, blocks, label, tol, rows):
    ok = True
    D = (A_ref - A).tocsr()
    for rn, _, ro, rs in blocks:          # rows = states (transposed Jacobian)
        for cn, _, co, cs in blocks:      # cols = residuals
            ref = A_ref[ro : ro + rs][:, co : co + cs]
            nr = np.sqrt(ref.multiply(ref).sum())
            if nr == 0:
                continue
            d = D[ro : ro + rs][:, co : co + cs]
            e = np.sqrt(d.multiply(d).sum()) / nr
            rows.append((f"{label}[{rn} x {cn}Res]", e, e <= tol))
            ok &= e <= tol
    pat_only_ref = ((A_ref != 0).astype(np.int8) - (A != 0).astype(np.int8))
    rows.append((f"{label} entries only in reference / only here", (int((pat_only_ref > 0).sum()), int((pat_only_ref < 0).sum())), True))
    return ok


def compare(case_dir, dump_dir, time="0", function="CD", engine="gpu", tol=1e-6, solver="DASimpleFoam", adj_time=None,
            norm_states=None, verbose=True):
    from dafoam_amd.meshgen import _InputGeometry, wall_distance

    mesh = foam_io.read_polymesh(case_dir)
    g = _InputGeometry(mesh)
    case = foam_io.read_case(case_dir, solver_name=solver, time=time, y_wall=wall_distance(mesh, g.C, g.Cf, g.Sf))
    norm_states = norm_states or {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0, "T": 300.0}
    E = Engine(case, engine, norm_states)
    blocks = state_blocks(case)
    n = sum(b[3] for b in blocks)
    rows, ok = [], True
    mats = {}
    for isPC, fn in ((0, "dRdWT.bin"), (1, "dRdWTPC.bin")):
        path = os.path.join(dump_dir, fn)
        if not os.path.exists(path):
            rows.append((fn, "absent", True))
            continue
        A_ref = petsc_io.read_mat(path).tocsr()
        assert A_ref.shape == (n, n), f"{fn}: {A_ref.shape} vs {n} states (single rank, adjStateOrdering state expected)"
        A = E.matrix(isPC)
        mats[isPC] = (A_ref, A)
        ok &= block_report(A_ref, A, blocks, fn[:-4], tol if not isPC else max(tol, 1e-5), rows)  # the PC is a finite-difference matrix
    if 0 in mats:
        A_ref, A = mats[0]
        for nm, v in (("0.001*1", np.full(n, 1e-3)), ("random", np.random.default_rng(0).standard_normal(n))):
            a, b = A_ref @ v, A @ v
            e = np.linalg.norm(a - b) / np.linalg.norm(a)
            rows.append((f"dRdW^T.v (v = {nm})", e, e <= tol))
            ok &= e <= tol
        rows.append(("||dRdW^T (0.001*1)||_2 (runUnitTests_DATurbModel.py:96 convention)", float(np.linalg.norm(A @ np.full(n, 1e-3))), True))
        nj = os.path.join(dump_dir, "norms.json")
        if os.path.exists(nj):
            known = json.load(open(nj)).get("dRdWTv_0.001")
            if known:
                e = abs(np.linalg.norm(A @ np.full(n, 1e-3)) - known) / abs(known)
                rows.append(("known answer ||dRdW^T (0.001*1)||", e, e <= tol))
                ok &= e <= tol
    colf = os.path.join(dump_dir, "dRdWColoring_1.bin")
    if os.path.exists(colf):
        col = np.rint(petsc_io.read_vec(colf)).astype(np.int64)
        con = sp.csr_matrix(E.connectivity())
        # DAColoring::validateColoring (DAColoring.C:931-1037): no residual row holds two columns of one colour
        bad = 0
        for i in range(con.shape[0]):
            c = col[con.indices[con.indptr[i] : con.indptr[i + 1]]]
            bad += c.size - np.unique(c).size
        rows.append((f"reference colouring ({int(col.max()) + 1} colours) conflicts on this repo's dRdWCon", bad, bad == 0))
        ok &= bad == 0
    adj_time = adj_time if adj_time is not None else time
    if os.path.exists(os.path.join(case_dir, str(adj_time), f"adjoint_{function}_U")) and 0 in mats:
        psi_ref = read_adjoint_fields(case_dir, adj_time, function, case)
        rhs = mats[0][0] @ psi_ref  # the right-hand side the reference solved, recovered from its own operator
        psi, fail = E.solve(rhs)
        for nm, _, off, size in blocks:
            e = np.linalg.norm(psi[off : off + size] - psi_ref[off : off + size]) / max(np.linalg.norm(psi_ref[off : off + size]), 1e-300)
            rows.append((f"psi[{nm}]", e, e <= tol))
            ok &= e <= tol
        rows.append(("adjoint solve fail flag", int(fail), fail == 0))
        ok &= fail == 0
    if verbose:
        for name, val, passed in rows:
            sval = f"{val:.3e}" if isinstance(val, float) else str(val)
            print(f"{'PASS' if passed else 'FAIL'}  {name:70s} {sval}")
        print("VERDICT:", "parity within tolerance" if ok else "MISMATCH", f"(engine {engine}, tol {tol:g})")
    return ok, rows


def write_self_dump(work, engine="oracle", dims=(6, 5, 4)):
    """This repo's own dumps in the reference's on-disk formats (synthetic channel at a converged oracle primal)."""
    from dafoam_amd.meshgen import channel_case

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import NORM_STATES
    from oracle.foam_mesh import Geometry
    from oracle.primal import solve_primal

    case = channel_case(*dims, perturb=0.0, lengths=(1.0, 0.2, 0.2), grading_y=2.0)
    W, _ = solve_primal(case, Geometry(case.mesh), max_iters=600, tol=1e-10)
    case.states = W
    case_dir = os.path.join(work, "case")
    foam_io.write_case(case_dir, case)
    E = Engine(foam_io.read_case(case_dir, y_wall=case.y_wall), engine, NORM_STATES)
    case_r = E.case
    A, P = E.matrix(0), E.matrix(1)
    petsc_io.write_mat(os.path.join(work, "dRdWT.bin"), A)
    petsc_io.write_mat(os.path.join(work, "dRdWTPC.bin"), P)
    if engine == "gpu":
        col = E.D.solver.getColoring()[0]
    else:
        col = E.col
    petsc_io.write_vec(os.path.join(work, "dRdWColoring_1.bin"), np.asarray(col, dtype=float))
    n = A.shape[0]
    N = case.mesh.n_cells
    rhs = np.zeros(n)
    rhs[0 : 3 * N : 3] = 1.0
    psi, _ = E.solve(rhs)
    foam_io.write_adjoint_fields(case_dir, case_r, "CD", "0", psi, state_blocks(case_r))
    json.dump({"dRdWTv_0.001": float(np.linalg.norm(A @ np.full(n, 1e-3)))}, open(os.path.join(work, "norms.json"), "w"))
    return case_dir


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--case")
    ap.add_argument("--dump")
    ap.add_argument("--time", default="0")
    ap.add_argument("--adj-time", default=None)
    ap.add_argument("--function", default="CD")
    ap.add_argument("--engine", default="gpu", choices=["gpu", "oracle"])
    ap.add_argument("--solver", default="DASimpleFoam")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.self_test:
        import tempfile

        with tempfile.TemporaryDirectory() as work:
            case_dir = write_self_dump(work, engine=a.engine)
            ok, _ = compare(case_dir, work, engine=a.engine, tol=a.tol)
        raise SystemExit(0 if ok else 1)
    if not a.case:
        ap.error("--case is required (or --self-test)")
    ok, _ = compare(a.case, a.dump or a.case, time=a.time, function=a.function, engine=a.engine, tol=a.tol, solver=a.solver, adj_time=a.adj_time)
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
