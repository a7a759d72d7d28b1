#!/usr/bin/env python
"""bench.py - adjoint hot-path benchmark (BASELINE.json metric: adjoint GMRES iterations/s + dRdWTPsi GB/s).

Workload (default `--workload naca`, at EVERY N): BASELINE configs[2] - DASimpleFoam + SA on a 2.016 M-cell NACA0012 wing section: the
O-grid section of --naca 200 x 63 cells (first cell 4e-5 chords, far field 20 chords) extruded to 160 spanwise layers of 0.025 chords
between symmetry planes, full GMRES adjoint, linearised about a CONVERGED primal: the flow is solved first by this library's
Newton-Krylov primal with grid sequencing on the section, extrusion and a Newton polish on the wing (dafoam_amd/workloads.py; untimed
set-up, reported).  The 800 x 250 section BASELINE.md names converges as a primal but its adjoint needs > 1000 iterations.
N > 1: the SAME wing about the SAME primal (converged on rank 0), cut into N spanwise slabs, ONE global solve ("scaling": "strong").
`--workload channel` is the round-1..3 bump channel (weak scaling, or --global-cells for a fixed global size).
No preconditioner option is passed: amd.pcUpwindBlend 0.5, the deflated coarse mode and the compressed Krylov basis are the LIBRARY's
defaults (config.pc_options_passed_by_bench lists what the command line overrode: normally nothing).
One "step" = one right-preconditioned GMRES iteration of the adjoint solve: node-block ILU(0) apply (two sync-free triangular
sweeps) + coarse correction + dRdW^T.z SpMV + orthogonalisation against the j basis vectors + norm, on the device-resident system
assembled by coloured dual-number / FD perturbation of the HIP residual.  Matrices, rhs and Krylov basis are resident in HBM.

Order of events: set-up -> the FULL solve to gmresRelTol = 1e-6 with the reference's defaults (gmresRestart = gmresMaxIters = 1000;
`config.solve`: iterations, time_to_tolerance_s, fail flag of DALinearEqn.C:422-434) -> the TIMED WINDOW of the driver contract:
a second solve of the same system is advanced untimed to the MEAN basis depth of the full solve (at least --warmup iterations),
then EXACTLY K (= --steps) iterations are timed.  The orthogonalisation cost grows linearly with the basis depth, so the window
rate is the mean rate of the whole solve (config.solve.iterations_per_sec_whole_solve is printed beside it; rounds 1-3 timed the window
[W, W + K) at small depths - `--window-at-warmup` restores that, config.value_is says which one a line carries).

  python bench.py --gpus 1 --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0) with `roofline` (dRdW^T.psi SpMV, HIP-event timed on the launch stream), `roofline_pc`,
`roofline_iteration` and `cpu_baseline`: the oracle's OpenMP C port (oracle/csrc/oracle_krylov_omp.c; threads = the container's CPU
quota) iterating on the SAME operator and PC matrix at the bench size (copied back from the device; the sample sits at basis depths
j < ~35, the GPU window at the solve's mean depth), plus `psi_parity_200k`: the 198 k-cell wing of the same family solved to
--parity-tol (1e-9) by the GPU path and by an INDEPENDENT host pipeline that assembles its own dRdW^T / dRdWTPC
(oracle/adjoint_host.py) - |psi_gpu - psi_cpu| / |psi_cpu| and the host's Jacobian-build time beside the GPU's.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NORM = {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0}  # reference tests/runRegTests_AeroOpt.py:83
NORM_RHO = {"U": 50.0, "p": 1.0e5, "T": 300.0, "nuTilda": 1e-3, "phi": 1.0}  # compressible solvers (the tier's NORM_STATES_RHO)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--nx", type=int, default=int(os.environ.get("DAS_BENCH_NX", 250)))
    ap.add_argument("--ny", type=int, default=int(os.environ.get("DAS_BENCH_NY", 100)))
    ap.add_argument("--nz", type=int, default=int(os.environ.get("DAS_BENCH_NZ", 80)))
    ap.add_argument("--workload", default=os.environ.get("DAS_BENCH_WORKLOAD", "naca"),
                    help="naca (default at N = 1): NACA0012 O-grid of --naca n_around n_normal nz cells, extruded in span, linearised about the "
                         "primal converged on the GPU by grid sequencing; channel: nx x ny x nz bump channel (state prolonged from a converged "
                         "coarse primal; the N > 1 workload)")
    ap.add_argument("--global-cells", type=int, default=int(os.environ.get("DAS_BENCH_GLOBAL_CELLS", 0)),
                    help="STRONG scaling (channel workload): ONE fixed global mesh of about this many cells (nx chosen as a multiple of the rank count, ny x nz "
                         "kept) cut into N slabs - e.g. 10000000 for the north-star 10 M-cell case at 2/4/8 GPUs; 0 (default): weak scaling, nx x ny x nz cells per GPU")
    ap.add_argument("--naca-dz", type=float, default=0.025, help="naca: spanwise layer thickness (chords); 160 layers x 0.025 = a wing section of aspect ratio 4 between symmetry planes")
    ap.add_argument("--naca-synthetic", action="store_true", help="naca: the round-3 synthetic noisy boundary-layer state instead of the converged primal")
    ap.add_argument("--window-at-warmup", action="store_true", help="time the K steps at basis sizes [W, W+K) instead of around the mean depth of the full solve")
    ap.add_argument("--parity-tol", type=float, default=1e-9, help="relative residual both sides of the psi parity leg are solved to (bar on the psi difference: 1e-6)")
    ap.add_argument("--no-parity", action="store_true", help="skip the 200 k-cell psi parity leg (GPU vs all-core CPU port)")
    ap.add_argument("--pc-blend", type=float, default=None, help="amd.pcUpwindBlend: weight of the second-order (linearUpwindV) correction in the PC residual "
                         "(the reference user's choice of div(pc) in fvSchemes); default: the library's (0.5)")
    ap.add_argument("--deflation", type=int, default=0,
                    help="amd.gmresDeflation k > 0: the full solve runs GMRES with deflated restarting (basis = --solve-restart vectors, k harmonic Ritz vectors kept); "
                         "opt-in, not yet measured on the device (DESIGN.md section 10 item 0b); the timed window stays the undeflated iteration at the solve's mean basis depth")
    ap.add_argument("--ordering", default=os.environ.get("DAS_BENCH_ORDERING", "rcm"), help="adjEqnOption.jacMatReOrdering: rcm | natural")
    ap.add_argument("--naca", type=int, nargs=3, default=[200, 63, 160], help="naca: cells around the section, wall-normal, spanwise layers")
    ap.add_argument("--naca-first-cell", type=float, default=4.0e-5, help="naca: first cell height (chords) of the section")
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("DAS_BENCH_CPU_SECONDS", 15.0)))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the solve-to-tolerance phase")
    ap.add_argument("--pctype", default=os.environ.get("DAS_BENCH_PCTYPE", "bilu"))
    ap.add_argument("--fp32-factor", type=int, default=int(os.environ.get("DAS_BENCH_PCFP32", 0)))
    ap.add_argument("--krylov-gb", type=float, default=float(os.environ.get("DAS_BENCH_KRYLOV_GB", 160.0)))
    ap.add_argument("--solve-restart", type=int, default=1000)
    ap.add_argument("--solve-rtol", type=float, default=1e-6, help="gmresRelTol of the solve to tolerance (default: the reference's 1e-6, pyDAFoam.py:526-548)")
    ap.add_argument("--solve-maxit", type=int, default=1000)
    ap.add_argument("--rho-levels", type=int, default=1, help="DARhoSimpleFoam with --converge-primal: grid-sequencing levels of the primal (2: converge nx/2 x ny/2 x nz/2 first, prolong)")
    ap.add_argument("--converge-primal", action="store_true", help="converge the flow state with the GPU Newton-Krylov primal before the adjoint (opt-in: the adjoint's conditioning does not depend on it, DESIGN.md section 6b)")
    ap.add_argument("--coarse-agg", type=int, default=int(os.environ.get("DAS_BENCH_COARSE", -1)), help="two-level PC: aggregates (-1 auto, 0 off)")
    ap.add_argument("--coarse-mode", default=os.environ.get("DAS_BENCH_COARSE_MODE"), help="amd.pcCoarseMode additive | deflated (default: the library's, deflated)")
    ap.add_argument("--orth", default=os.environ.get("DAS_BENCH_ORTH", "dcgs2"), help="dcgs2 (delayed re-orthogonalisation, 2 basis reads / iteration) | cgs (reference: refine if needed)")
    ap.add_argument("--naca-sweep", type=float, default=0.0, help="naca: sweep angle in degrees (round 6: a genuinely 3-D wing segment - every layer its own section; with --naca-taper)")
    ap.add_argument("--naca-taper", type=float, default=0.0, help="naca: fraction of the chord lost from the first to the last layer (0.3: tip chord 0.7)")
    ap.add_argument("--naca-polish-steps", type=int, default=None, help="naca: Newton steps on the 3-D mesh (default 3 for the extruded section, at most 60 for a swept / tapered segment)")
    ap.add_argument("--naca-fold", dest="naca_fold", action="store_true", default=True,
                    help="naca (default): number the cells around the section 0, n-1, 1, n-2, ... - the two sides of the O-grid's seam become neighbours in the numbering, "
                         "like a renumberMesh pass; the index-ordered first-fit colouring is then no longer serialised ring after ring (2 M cells: 415 colours instead of the "
                         "538 of the speculative fallback; same iteration counts)")
    ap.add_argument("--naca-no-fold", dest="naca_fold", action="store_false", help="naca: the plain numbering 0, 1, ..., n-1 around the section (rounds 3-5)")
    ap.add_argument("--naca-state-cache", default=None, metavar="FILE.npy", help="naca: load the converged 3-D state from this file if it exists (same mesh arguments!), else converge and save it - "
                    "repeated experiments on one box skip the primal")
    ap.add_argument("--naca-partition", default="columns", choices=["columns", "span", "around"],
                    help="naca, N > 1: 'columns' (default) = blocks in the (around, wall-normal) index plane, every rank keeps whole spanwise columns of cells - the cut "
                         "never crosses the strong spanwise coupling of the thin layers; 'around' = sectors around the airfoil; 'span' = spanwise slabs of whole layers")
    ap.add_argument("--asm-overlap", type=int, default=None, help="adjEqnOption.asmOverlap (N > 1: rings of ghost cells in every rank's sub-domain solve); default: the library's (1, the reference's)")
    ap.add_argument("--solver", default="DASimpleFoam", choices=["DASimpleFoam", "DARhoSimpleFoam", "DATurboFoam"],
                    help="BASELINE configs[3] / [4]: the compressible solvers run on the bump channel of --nx/--ny/--nz cells per GPU with a synthetic subsonic state "
                         "(p 101325, T 300; DATurboFoam: one MRF zone, rotating hub) - N > 1: RCB cell partition of the global channel (ShardedAdjointGeneral.scattered); "
                         "no converged primal and no psi parity leg for them (the host adjoint of the parity leg covers DASimpleFoam)")
    ap.add_argument("--dump-psi", default=None, metavar="FILE.npy", help="after the solve to tolerance: rank 0 writes psi in the GLOBAL state ordering (owned entries of every rank gathered, "
                    "face states back in the global face orientation) - psi of an N-rank run against psi of the 1-rank run")
    ap.add_argument("--amd", action="append", default=[], metavar="KEY=VALUE", help="experiments: any amd.* option, e.g. --amd gradFaceParallel=0 (listed in config.pc_options_passed_by_bench)")
    return ap.parse_args()


def pmc_traffic(op_nnz, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json), if they were taken on this
    workload (same operator nnz); None otherwise - counters cannot be collected inside a timed run."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for w in json.load(f)["workloads"]:
                if int(w["dRdWT_nnz"]) == int(op_nnz) and kernel in w:
                    return float(w[kernel])
    except (OSError, ValueError, KeyError):
        pass
    return None


_T0 = time.time()


def stage(msg):
    """Progress on stderr (DAS_BENCH_VERBOSE=1): where the time of a run goes; never on stdout (ONE JSON line there)."""
    if os.environ.get("DAS_BENCH_VERBOSE"):
        print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def make_opts(a, dev_index, restart, maxit, rtol):
    return {
        "solverName": a.solver,
        "normalizeStates": dict(NORM if a.solver == "DASimpleFoam" else NORM_RHO),
        "adjEqnOption": dict({"gmresRestart": int(restart), "gmresMaxIters": int(maxit), "gmresRelTol": rtol, "gmresAbsTol": 1e-300, "printInfo": 0,
                              "jacMatReOrdering": a.ordering}, **({"asmOverlap": int(a.asm_overlap)} if a.asm_overlap is not None else {})),
        # amd.*: ONLY what the command line / the device budget asks for explicitly - the preconditioner is the library's own default
        # (round 5: amd.pcUpwindBlend 0.5 + deflated coarse mode are library defaults, no longer bench switches)
        "amd": dict({"maxKrylovBytes": int(a.krylov_gb * 2**30)},
                    **({"pcType": a.pctype} if a.pctype != "bilu" else {}), **({"pcFactorFP32": a.fp32_factor} if a.fp32_factor else {}),
                    **({"pcCoarseAggregates": a.coarse_agg} if a.coarse_agg != -1 else {}), **({"pcCoarseMode": a.coarse_mode} if a.coarse_mode else {}),
                    **({"gmresOrthogonalization": a.orth} if a.orth != "dcgs2" else {}), **({"pcUpwindBlend": float(a.pc_blend)} if a.pc_blend is not None else {}),
                    # compressible solvers with --converge-primal: the cold-start settings that converge the bump channel from its smooth synthetic state
                    # (round 6, profiles/r08d_*: CFL ramp from tau 0.1, growth 1.3, PC rebuilt every step, pseudo-time term on all rows: 6.6e7 -> 5e-2 in
                    # 35 Newton steps at 100 k cells; switched evolution relaxation diverges there, the NACA cold-start settings stall)
                    **({"primalTauMode": "ramp", "primalTau0": 0.1, "primalTauGrowth": 1.3, "primalPCLag": 1} if (a.solver != "DASimpleFoam" and getattr(a, "converge_primal", False)) else {}),
                    **_amd_overrides(a)),
        "amdDevice": dev_index,
    }


def compressible_channel(a, nx):
    """DARhoSimpleFoam / DATurboFoam on the bump channel (the generators of the GPU tier: perfect gas, p 101325, T 300, subsonic; DATurboFoam with
    one MRF zone and a rotating hub - reference tests/runRegTests_DARhoSimpleFoam*.py, runRegTests_DATurboFoam*.py), synthetic smooth state."""
    from dafoam_amd.meshgen import rho_channel_case, turbo_channel_case

    kw = dict(lengths=(2.0, 0.2, 0.2), grading_y=2.0)
    if getattr(a, "converge_primal", False):
        kw["perturb"] = 0.0  # a primal solve starts from the smooth guess, not from the seeded noise of the rate-only runs
    return rho_channel_case(nx, a.ny, a.nz, **kw) if a.solver == "DARhoSimpleFoam" else turbo_channel_case(nx, a.ny, a.nz, **kw)


def naca_partition(cid, dims, world, kind, fold=False):
    """Cell partition of the extruded O-grid (cell id = i + n_around (j + n_normal k)).  The spanwise layers are thin (0.025 chords) against the
    in-plane size of most cells, so the spanwise faces carry the strongest couplings of the wing's Jacobian: 'columns' and 'around' keep
    every spanwise column of cells on one rank (the cut crosses only in-plane faces), 'span' cuts exactly those couplings (round 5:
    1 rank 641 iterations, 2 spanwise slabs > 1000)."""
    na, nn, nz = dims
    from dafoam_amd.meshgen import naca_ring_position

    i, j, k = naca_ring_position(cid, na, fold), (cid // na) % nn, cid // (na * nn)
    if kind == "span":
        return (k * world // nz).astype(np.int32)
    if kind == "around":
        return (i * world // na).astype(np.int32)
    # columns: recursive bisection of the (i, j) index rectangle, always across its longer side (in cells)
    part = np.zeros(cid.size, dtype=np.int32)

    def rec(sel, lo_i, hi_i, lo_j, hi_j, p0, np_):
        if np_ == 1:
            part[sel] = p0
            return
        nl = np_ // 2
        if (hi_i - lo_i) >= (hi_j - lo_j):
            mid = lo_i + (hi_i - lo_i) * nl // np_
            left = i[sel] < mid
            rec(sel[left], lo_i, mid, lo_j, hi_j, p0, nl)
            rec(sel[~left], mid, hi_i, lo_j, hi_j, p0 + nl, np_ - nl)
        else:
            mid = lo_j + (hi_j - lo_j) * nl // np_
            left = j[sel] < mid
            rec(sel[left], lo_i, hi_i, lo_j, mid, p0, nl)
            rec(sel[~left], lo_i, hi_i, mid, hi_j, p0 + nl, np_ - nl)

    rec(np.arange(cid.size), 0, na, 0, nn, 0, world)
    return part


def _wing3d_kwargs(a):
    """naca_extruded_case arguments of a swept / tapered wing segment (--naca-sweep / --naca-taper): the mesh transformation and enough
    Newton steps to converge the primal on it (the extruded section's state is only a starting guess there)."""
    swept = bool(a.naca_sweep or a.naca_taper)
    kw = {}
    if swept or a.naca_fold:
        kw["case_kwargs"] = dict(({"sweep_deg": float(a.naca_sweep), "taper": float(a.naca_taper)} if swept else {}), **({"fold_seam": True} if a.naca_fold else {}))
    steps = a.naca_polish_steps if a.naca_polish_steps is not None else (60 if swept else 3)
    kw["polish_steps"] = int(steps)
    if swept:
        kw["polish_tol"] = 1e-10  # relative to the residual of the section's state on the deformed mesh (1e5): the extruded wing's level, 1e-5
    return kw


def _amd_overrides(a):
    out = {}
    for kv in a.amd:
        k, v = kv.split("=", 1)
        try:
            out[k] = int(v)
        except ValueError:
            try:
                out[k] = float(v)
            except ValueError:
                out[k] = v
    return out


def main():
    a = parse()
    if os.environ.get("DAS_BENCH_FAULT_DUMP"):
        # diagnosis of a stalled multi-rank run: every rank dumps its Python stacks to stderr after this many seconds (and keeps running)
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["DAS_BENCH_FAULT_DUMP"]), repeat=False, exit=False)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # debugging aids for boxes with a single GPU (never set by the driver): all ranks on device 0, host-staged gloo
    backend = os.environ.get("DAS_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DAS_BENCH_ONE_GPU") == "1" else local_rank
    if os.environ.get("DAS_BENCH_ONE_GPU") == "1" and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # several ranks share ONE device (the 1-GPU test box): the per-XCD ticket counters of the preconditioner sweeps assume that a launch has
        # workgroups resident on every XCD - true for one process per GPU, not when eight processes compete for the compute units (round 6: an 8-rank
        # run at 2 M cells stalled with one rank's stream never finishing).  The device-wide counter needs no residency assumption.
        os.environ.setdefault("DAS_BILU_XCD", "0")
    torch.cuda.set_device(dev_index)
    if world > 1:
        import datetime

        # a bounded collective timeout: a peer that never arrives aborts the job instead of hanging the node
        tmo = datetime.timedelta(seconds=int(os.environ.get("DAS_BENCH_COLLECTIVE_TIMEOUT", 900)))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        # host-side setup is OpenMP-parallel: give every rank its share of the cores (torch.distributed.run exports
        # OMP_NUM_THREADS=1 when it is unset; the library reads it when it is loaded below)
        os.environ["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // world))

    import __graft_entry__ as ge

    # the .so files travel prebuilt; if they have to be (re)built, one rank per node does it
    if local_rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    if local_rank != 0:
        ge.build()
    from dafoam_amd import _capi
    from dafoam_amd.meshgen import bench_channel_case
    from dafoam_amd.pyDAFoam import PYDAFOAM
    from dafoam_amd.pyDASolvers import KSP, Mat

    L = _capi.lib()
    t_setup = time.time()
    if a.global_cells > 0 or a.solver != "DASimpleFoam":
        a.workload = "channel"  # --global-cells: the structured channel cut into slabs; the compressible solvers: the channel generators
    opts = make_opts(a, dev_index, a.solve_restart, a.solve_maxit, 1e-6)
    sharded = None
    primal, case2d = None, None
    if world > 1 and a.solver != "DASimpleFoam":
        # BASELINE configs[3] / [4]: compressible solver on a cell-partitioned mesh - rank 0 generates the global channel (nx x world columns),
        # an RCB partition cuts it, the extended sub-meshes are scattered (the reference's decomposePar in memory)
        from dafoam_amd.distributed import ShardedAdjointGeneral, rcb_partition

        gcase, part = None, None
        if rank == 0:
            gcase = compressible_channel(a, a.nx * world)
            from dafoam_amd.meshgen import _InputGeometry

            part = rcb_partition(_InputGeometry(gcase.mesh).C, world)
            stage(f"rank 0: global {a.solver} channel ready ({gcase.mesh.n_cells} cells), scattering {world} sub-meshes")
        sharded = ShardedAdjointGeneral.scattered(gcase, part, opts, device_index=dev_index, src=0)
        del gcase
        D = sharded.D
        case = sharded.case
        ncell = int(sharded.owned[3 * case.mesh.n_cells : 4 * case.mesh.n_cells].sum())
    elif world > 1 and a.workload == "naca":
        # N > 1 keeps the N = 1 workload: the SAME wing, linearised about the SAME converged primal, cut into N
        # spanwise slabs of whole cell layers (strong scaling).  Rank 0 converges the primal on its GPU exactly like the N = 1 run
        # (section by grid sequencing, extrusion, Newton polish), extracts every rank's extended sub-mesh (owned cells + 3 ghost rings)
        # and scatters them - the reference's decomposePar step done in memory (pyDAFoam.py:597-604)
        from dafoam_amd.distributed import ShardedAdjointGeneral

        gcase, part = None, None
        if rank == 0:
            from dafoam_amd.workloads import naca_converged_primal, naca_extruded_case

            t0 = time.time()
            if a.naca_synthetic:
                from dafoam_amd.meshgen import naca0012_case

                gcase = naca0012_case(a.naca[0], a.naca[1], a.naca[2], span=a.naca_dz * a.naca[2], first_cell=a.naca_first_cell, fold_seam=a.naca_fold)
            else:
                case2d, lv = naca_converged_primal(a.naca[0], a.naca[1], options=opts, first_cell=a.naca_first_cell, verbose=bool(os.environ.get("DAS_BENCH_VERBOSE")),
                                                   case_kwargs=({"fold_seam": True} if a.naca_fold else None))
                gcase, ex = naca_extruded_case(case2d, (a.naca[0], a.naca[1]), a.naca[2], dz=a.naca_dz, first_cell=a.naca_first_cell, options=opts,
                                               verbose=bool(os.environ.get("DAS_BENCH_VERBOSE")), **_wing3d_kwargs(a))
                primal = {"method": "rank 0: pseudo-transient Newton-Krylov (das_solve_primal), grid sequencing, spanwise extrusion, Newton polish; then scattered",
                          "levels": [{k: (list(v) if isinstance(v, tuple) else v) for k, v in r.items()} for r in lv],
                          "extruded": {k: (list(v) if isinstance(v, tuple) else v) for k, v in ex.items()}, "seconds": time.time() - t0}
            # spanwise slabs of whole layers (the generator numbers the cells layer by layer: layer = cell // (n_around * n_normal))
            cid = np.arange(gcase.mesh.n_cells, dtype=np.int64)
            part = naca_partition(cid, a.naca, world, a.naca_partition, fold=a.naca_fold)
            stage(f"rank 0: global wing ready ({gcase.mesh.n_cells} cells), scattering {world} sub-meshes")
        sharded = ShardedAdjointGeneral.scattered(gcase, part, opts, device_index=dev_index, src=0)
        del gcase
        D = sharded.D
        case = sharded.case
        ncell = int(sharded.owned[3 * case.mesh.n_cells : 4 * case.mesh.n_cells].sum())  # owned cells of this rank (p block of the owned mask)
    elif world > 1:
        # weak scaling: the global channel has nx*world cell columns, every rank owns nx of them (+3 ghost layers)
        from dafoam_amd.distributed import ShardedAdjoint

        if a.global_cells > 0:  # strong scaling: the global mesh is fixed, every rank owns NX / world cell columns
            a.nx = max(1, int(round(a.global_cells / float(a.ny * a.nz * world))))
        sharded = ShardedAdjoint(a.nx * world, a.ny, a.nz, opts, device_index=dev_index)
        D = sharded.D
        case = sharded.case
        ncell = a.nx * a.ny * a.nz
    else:
        if a.global_cells > 0:
            a.nx = max(1, int(round(a.global_cells / float(a.ny * a.nz))))
        if a.workload == "naca" and not a.naca_synthetic:
            # the reference linearises about a converged primal (mphys_dafoam.py:314-433): Newton-Krylov primal on the GPU, grid
            # sequencing on the one-layer O-grid, spanwise extrusion + a few Newton steps on the extruded mesh
            from dafoam_amd.workloads import naca_converged_primal, naca_extruded_case

            t0 = time.time()
            case2d, lv = naca_converged_primal(a.naca[0], a.naca[1], options=opts, first_cell=a.naca_first_cell, verbose=bool(os.environ.get("DAS_BENCH_VERBOSE")),
                                                   case_kwargs=({"fold_seam": True} if a.naca_fold else None))
            t2d = time.time() - t0
            if a.naca[2] > 1 and a.naca_state_cache and os.path.exists(a.naca_state_cache):
                kw3 = _wing3d_kwargs(a)
                kw3["polish_steps"] = 0
                case, ex = naca_extruded_case(case2d, (a.naca[0], a.naca[1]), a.naca[2], dz=a.naca_dz, first_cell=a.naca_first_cell, options=opts, **kw3)
                case.states = np.load(a.naca_state_cache)
                ex["loaded_from"] = a.naca_state_cache
            elif a.naca[2] > 1:
                case, ex = naca_extruded_case(case2d, (a.naca[0], a.naca[1]), a.naca[2], dz=a.naca_dz, first_cell=a.naca_first_cell, options=opts,
                                              verbose=bool(os.environ.get("DAS_BENCH_VERBOSE")), **_wing3d_kwargs(a))
                if a.naca_state_cache:
                    np.save(a.naca_state_cache, np.asarray(case.states))
            else:
                case, ex = case2d, None
            primal = {"method": "pseudo-transient Newton-Krylov (das_solve_primal), grid sequencing on the one-layer O-grid, spanwise extrusion, Newton polish",
                      "levels": [{k: (list(v) if isinstance(v, tuple) else v) for k, v in r.items()} for r in lv], "seconds_2d": t2d,
                      "extruded": ({k: (list(v) if isinstance(v, tuple) else v) for k, v in ex.items()} if ex else None),
                      "seconds": time.time() - t0}
        elif a.workload == "naca":
            from dafoam_amd.meshgen import naca0012_case

            case = naca0012_case(a.naca[0], a.naca[1], a.naca[2], span=a.naca_dz * a.naca[2], first_cell=a.naca_first_cell, fold_seam=a.naca_fold)
        elif a.solver == "DARhoSimpleFoam" and a.converge_primal and a.rho_levels > 1:
            # BASELINE configs[3] about a CONVERGED compressible primal: grid sequencing on the device (round 6)
            from dafoam_amd.workloads import rho_channel_converged_primal

            t0 = time.time()
            case, lv = rho_channel_converged_primal(a.nx, a.ny, a.nz, options=opts, levels=a.rho_levels, verbose=bool(os.environ.get("DAS_BENCH_VERBOSE")),
                                                    case_kwargs=dict(lengths=(2.0, 0.2, 0.2), grading_y=2.0))
            primal = {"method": "pseudo-transient Newton-Krylov (das_solve_primal), grid sequencing on the bump channel: cold start by a CFL ramp on the coarsest level, "
                                "prolonged starts with switched evolution relaxation", "levels": [{k: (list(v) if isinstance(v, tuple) else v) for k, v in r.items()} for r in lv],
                      "seconds": time.time() - t0, "fail": int(lv[-1]["fail"]), "res0": lv[-1]["res0"], "res": lv[-1]["res"], "steps": lv[-1]["steps"]}
        elif a.solver != "DASimpleFoam":
            case = compressible_channel(a, a.nx)
        else:
            # state = prolongation of a converged coarse primal (dafoam_amd/data/channel_primal_coarse.npz)
            case = bench_channel_case(a.nx, a.ny, a.nz)
        ncell = case.mesh.n_cells
        D = PYDAFOAM(options=opts, case=case)
    t_case = time.time() - t_setup
    stage(f"case ready ({ncell} cells)")
    if a.converge_primal and world == 1 and primal is None:
        t0 = time.time()
        D.setOption("primalMinResTol", 1e-8) if hasattr(D, "setOption") else None
        pf = D.solvePrimal(maxSteps=100)
        primal = dict(D.primalInfo, fail=int(pf), seconds=time.time() - t0)
        primal["history"] = [float(v) for v in primal["history"]]
    h = D.solver._h
    n = D.getNLocalAdjointStates()
    R0 = np.zeros(n)
    D.solver.getResiduals(R0)
    if sharded is not None:
        # the norm of the GLOBAL residual: owned rows only (the ghost rows of an extended sub-mesh end at artificial cut patches - their
        # residuals mean nothing: rounds 5's N > 1 lines printed 3.9e4 / 9e-3 for a state whose owned rows are at 6e-7), summed over the ranks
        r2 = torch.tensor([float(np.sum(R0[sharded.owned] ** 2))], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(r2)
        primal_residual_norm = float(np.sqrt(r2.item()))
    else:
        primal_residual_norm = float(np.linalg.norm(R0))
    t0 = time.time()
    D.solver.runColoring()
    t_color = time.time() - t0
    _, ncolors = D.solver.getColoring()
    t0 = time.time()
    pc = Mat()
    D.solver.calcdRdWT(1, pc)
    t_pcmat = time.time() - t0
    ksp = KSP()
    t0 = time.time()
    D.solverAD.createMLRKSPMatrixFree(pc, ksp)
    global_coarse = 0
    if sharded is not None:
        # one pressure coarse space over all ranks (das_ksp_set_global_coarse) instead of one per rank
        sharded.pc, sharded.ksp = pc, ksp
        global_coarse = sharded.install_global_coarse()
    t_pc = time.time() - t0
    t0 = time.time()
    D.solverAD.initializedRdWTMatrixFree()
    t_op = time.time() - t0
    # rhs on the device (torch owns the buffers; the C-ABI gets raw pointers): the volume-averaged x-velocity functional,
    # dF/dW scaled like the reference scales its right-hand sides
    N = case.mesh.n_cells
    rhs_h = np.zeros(n)
    n_global = int(L.das_get_n_global_cells(h)) if sharded is not None else N
    rhs_h[0 : 3 * N : 3] = 1.0 / n_global
    if sharded is not None:
        rhs_h = np.where(sharded.owned, rhs_h, 0.0)
    rhs = torch.from_numpy(rhs_h).cuda()
    sol = torch.zeros(n, dtype=torch.float64, device="cuda")
    setup_s = time.time() - t_setup
    _est, _ord = C.c_double(-1.0), C.c_int(-1)
    L.das_ksp_get_pc_stability(ksp.handle, C.byref(_est), C.byref(_ord))
    pc_stab = {"estimate_max_abs_LUinv_P_e_minus_e": _est.value, "elimination_order": _ord.value}
    _so, _se = (C.c_int * 16)(), (C.c_double * 16)()
    _K = int(L.das_ksp_get_pc_subdomains(ksp.handle, _so, _se))
    pc_stab["subdomains_in_rank"] = _K
    if _K > 1:
        pc_stab["subdomain_elimination_orders"] = [int(_so[i]) for i in range(_K)]
        pc_stab["subdomain_estimates"] = [float(_se[i]) for i in range(_K)]
    stage(f"adjoint set-up done: colouring {t_color:.1f} s, dRdWTPC {t_pcmat:.1f} s, factorisation {t_pc:.1f} s, dRdWT {t_op:.1f} s")

    def check(rc):
        if rc < 0:
            raise RuntimeError(L.das_last_error().decode())
        return rc

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- solve to tolerance: the reference's defaults gmresRestart = gmresMaxIters = 1000 (pyDAFoam.py:526-548) --------------------
    # (N > 1: the same calls on every rank - ONE global solve, collective inside the library)
    solve = None
    r_eff = int(max(1, min(a.solve_restart, a.krylov_gb * 2**30 // (8 * n) - 2)))
    mean_depth = None
    if not a.no_solve:
        D.solver.updateDAOption({"adjEqnOption": {"gmresRestart": a.solve_restart, "gmresMaxIters": a.solve_maxit, "gmresRelTol": a.solve_rtol, "gmresAbsTol": 1e-14}})
        sol.zero_()
        barrier()
        t0 = time.perf_counter()
        if a.deflation > 0 and world == 1:
            # GMRES-DR through the host-vector entry (the deflated solver is not a begin / advance state machine)
            from dafoam_amd.pyDASolvers import Vec

            D.solver.updateDAOption({"amd": {"gmresDeflation": int(a.deflation)}})
            bvec, xvec = Vec(n), Vec(n)
            bvec.array[:] = rhs_h
            fail = D.solverAD.solveLinearEqn(ksp, bvec, xvec)
            D.solver.updateDAOption({"amd": {"gmresDeflation": 0}})
        else:
            check(L.das_ksp_begin_device(h, ksp.handle, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()), 0))
            while not check(L.das_ksp_advance(h, ksp.handle, 1000)):
                pass
            fail = check(L.das_ksp_end(h, ksp.handle))
        barrier()
        t_solve = time.perf_counter() - t0
        inf = ksp.info()
        hist = ksp.history()
        its = int(inf["iters"])
        mean_depth = float(np.mean(np.arange(its) % r_eff)) if its > 0 else 0.0
        if a.deflation > 0 and world == 1 and its > r_eff:  # later cycles run between depth k and m
            mean_depth = (r_eff * 0.5 * r_eff + (its - r_eff) * 0.5 * (a.deflation + r_eff)) / its
        binfo = ksp.basisInfo()
        solve = {"converged": fail == 0, "fail": int(fail), "iterations": its, "time_to_tolerance_s": t_solve,
                 "krylov_basis": {"storage": ("split: hi + lo floats per entry (8 B); the inner-product pass reads hi only, every vector-building pass hi + lo"
                                              if binfo["split"] else ("fp32" if binfo["fp32"] else "fp64")),
                                  "mapped_GB": binfo["mappedGB"], "bytes_per_vector": binfo["bytesPerVector"]},
                 "rel_residual": inf["res"] / inf["res0"] if inf["res0"] else None, "gmresRelTol": a.solve_rtol,
                 "gmresRestart": r_eff, "gmresMaxIters": a.solve_maxit, "mean_basis_depth": mean_depth, "gmresDeflation": int(a.deflation) if world == 1 else 0,
                 "rel_residual_at_1000_iterations": float(hist[1000] / hist[0]) if len(hist) > 1000 else None,
                 "rel_residual_every_100": [float(v / hist[0]) for v in hist[::100]],
                 "iterations_per_sec_whole_solve": its / t_solve}
        stage(f"solve: {its} iterations, {t_solve:.1f} s, fail {fail}")
        if a.dump_psi:
            x_h = sol.cpu().numpy()
            if sharded is not None and hasattr(sharded, "info"):
                own_m = sharded.owned
                mine = (sharded.key[own_m], x_h[own_m] * sharded.info["state_sign"][own_m])
                parts = [None] * world if rank == 0 else None
                dist.gather_object(mine, parts, dst=0)
                if rank == 0:
                    psi_g = np.full(sum(p_[0].size for p_ in parts), np.nan)
                    for kk, vv in parts:
                        psi_g[kk] = vv
                    np.save(a.dump_psi, psi_g)
            elif rank == 0:
                np.save(a.dump_psi, x_h)

    # ---- timed window (driver contract): W' untimed iterations, then EXACTLY K timed, inside ONE Arnoldi cycle.  W' = the mean basis
    # depth of the full solve minus K/2 (>= --warmup): the window rate is then the mean per-iteration rate of the whole solve ----------
    if mean_depth is None or a.window_at_warmup:
        j0 = int(a.warmup)
    else:
        j0 = int(max(a.warmup, min(round(mean_depth - 0.5 * a.steps), r_eff - a.steps - 1)))
    window_restart = max(j0 + a.steps, 1)
    # (amd.krylovBasisPrecision "auto" picks the basis storage from the orthogonalisation scheme and the basis SIZE, (restart + 2) n 8 B >= 1 GB ->
    #  split, not from the tolerance: the window's restart is >= the solve's mean depth, so at bench sizes the timed iterations run on the same
    #  storage type as the solve they stand for - config.solve.krylov_basis says which)
    D.solver.updateDAOption({"adjEqnOption": {"gmresRestart": window_restart, "gmresMaxIters": 10**9, "gmresRelTol": 1e-6, "gmresAbsTol": 1e-300}})
    sol.zero_()
    check(L.das_ksp_begin_device(h, ksp.handle, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()), 1))
    if j0 > 0:
        check(L.das_ksp_advance(h, ksp.handle, j0))
    L.das_timer_reset(h)
    L.das_timer_enable(h, 1)
    barrier()
    t0 = time.perf_counter()
    check(L.das_ksp_advance(h, ksp.handle, int(a.steps)))
    barrier()
    dt = time.perf_counter() - t0
    L.das_timer_enable(h, 0)
    check(L.das_ksp_end(h, ksp.handle))
    spmv_ms = L.das_timer_avg_ms(h, b"spmv")
    spmv_cnt = L.das_timer_count(h, b"spmv")
    pc_ms = L.das_timer_avg_ms(h, b"pc")
    win_info = ksp.info()
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- algorithmic bytes (SURVEY.md section 8d / BASELINE.md section 3) ---------------------------------------------
    opmat_nnz = int(L.das_get_con_nnz(h, 0))
    op_nnz = int(L.das_op_nnz(h))  # the operator drops exact zeros (jacLowerBounds 1e-30): its true nnz
    spmv_bytes = 12.0 * op_nnz + 4.0 * (n + 1) + 16.0 * n  # SURVEY.md 8(d) / BASELINE.md: the CSR formula, kept as THE algorithmic figure
    achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms and spmv_ms > 0 else None
    # what the operator's storage format really streams (vector rows packed: one column list + three value planes per group row)
    fmt_bytes = float(L.das_op_format_bytes(h)) + 16.0 * n
    fmt_GBps = fmt_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms and spmv_ms > 0 else None
    fac_entries = int(L.das_ksp_get_factor_nnz(ksp.handle))
    n_ext = int(L.das_ksp_get_n_ext(ksp.handle))
    pc_nnz = int(L.das_mat_nnz(pc.handle))
    if a.pctype == "bilu":
        # dense 8x8 node blocks: 8 (4) B per factor entry, one int32 per block, + b, y, z, out vectors
        pc_bytes = (4.0 if a.fp32_factor else 8.0) * fac_entries + 4.0 * fac_entries / 64.0 + 8.0 * (2 * n + 3 * n_ext)
    else:
        pc_bytes = 12.0 * fac_entries + 16.0 * n_ext
    pc_bytes_survey = 12.0 * pc_nnz + 16.0 * n  # B_pc = 12 nnz(L+U) + 16 n with the ILU(0) pattern of the PC matrix itself
    jmean = j0 + 0.5 * a.steps
    # the deflated two-level form (A-DEF1) applies the operator a second time inside every preconditioner apply
    products = (spmv_cnt / float(a.steps)) if a.steps else 1.0
    iter_bytes = spmv_bytes + pc_bytes + 32.0 * jmean * n + 48.0 * n  # BASELINE.md section 3 (CGS with refinement: 4 basis reads)
    # what this implementation has to move: the delayed re-orthogonalisation reads the basis twice per iteration
    orth = a.orth
    bi = ksp.basisInfo()
    basis_b = 4.0 if bi["fp32"] else 8.0
    # delayed re-orthogonalisation: one inner-product pass (4 B per entry with the split basis: hi only) + one update pass over the basis
    orth_bytes = ((4.0 if bi["split"] else basis_b) + basis_b) * jmean * n if orth == "dcgs2" else 4.0 * basis_b * jmean * n
    moved_bytes = products * spmv_bytes + pc_bytes + orth_bytes + 48.0 * n
    ms_step = dt / a.steps * 1e3
    stage(f"window: {a.steps} steps at depth {j0}: {ms_step:.2f} ms per step")

    out = None
    if rank == 0:
        cpu, parity = None, None
        if not a.no_cpu and world == 1:
            cpu = _with_deadline(lambda: cpu_port_at_bench_size(a, L, h, ksp, pc, n, N, op_nnz, pc_nnz, rhs_h, spmv_ms, pc_ms),
                                 float(os.environ.get("DAS_BENCH_CPU_DEADLINE", 150)), "cpu port at the bench size")
            stage(f"cpu port at the bench size: {({k: v for k, v in cpu.items() if k in ('value', 'ms_per_iteration', 'error', 'skipped')})}")
        if not a.no_parity and not a.no_cpu and world == 1 and a.solver != "DASimpleFoam":
            parity = {"skipped": "the host adjoint of the parity leg (oracle/adjoint_host.py) covers DASimpleFoam + SA; the compressible solvers' psi parity is in the GPU tier (small meshes, direct solves)"}
        elif not a.no_parity and not a.no_cpu and world == 1:
            parity = _with_deadline(lambda: psi_parity_200k(a, dev_index, case2d), float(os.environ.get("DAS_BENCH_PARITY_DEADLINE", 330)), "psi parity leg") \
                if not _OVERRUN else {"skipped": "the cpu port leg overran its deadline; the host is not usable for the CPU legs"}
            stage(f"psi parity leg: {({k: v for k, v in parity.items() if k in ('psi_rel_diff_gpu_vs_cpu', 'error')})}")
            if cpu is not None and parity is not None:
                cpu["psi_rel_diff_gpu_vs_cpu"] = parity.get("psi_rel_diff_gpu_vs_cpu")
                # the reference path INCLUDING the Jacobian build (north_star): measured on the 198 k-cell system of the parity leg, where the
                # host assembles its own matrices (at the bench size the host build would take ~10x as long: ~7 min on the container's quota)
                if isinstance(parity.get("cpu"), dict) and "jacobian_build_seconds" in parity["cpu"]:
                    cpu["jacobian_build_at_200k_cells"] = {"cpu_host_assembled_s": parity["cpu"]["jacobian_build_seconds"]["total"],
                                                          "gpu_s": parity["gpu"]["jacobian_build_seconds"]["total"], "cpu_threads": parity["cpu"]["threads"],
                                                          "cpu_solve_s": parity["cpu"]["seconds"], "gpu_solve_s": parity["gpu"]["seconds"]}
        pc_desc = ("node-block ILU(0) of FD dRdWTPC over the whole rank (8-slot cell nodes, 8x8 fp%s blocks), factorised on the device, "
                   "two sync-free sweeps per apply; + piecewise-constant pressure coarse space (%s)" % ("32" if a.fp32_factor else "64", D.getOption("amd")["pcCoarseMode"])) if a.pctype == "bilu" else \
            "RAS(overlap 1)+ILU(1) of FD dRdWTPC, RCB blocks of <= 1024 cells, one workgroup per block"
        out = {
            "metric": "adjoint_gmres_iterations_per_sec",
            # iterations/s of the ONE global adjoint solve all N ranks advance together (weak scaling: N x more cells per
            # iteration, so flat is ideal); config.cell_iterations_per_sec = global cells x iterations/s is the whole-job work rate
            "value": a.steps * 1.0 / dt,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            # naca: the SAME wing at every N (total work fixed); channel: nx x ny x nz cells per GPU (weak) unless --global-cells
            "scaling": "strong" if (a.global_cells > 0 or a.workload == "naca") else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ((f"BASELINE configs[{3 if a.solver == 'DARhoSimpleFoam' else 4}] family: {a.solver}+SA adjoint, {n_global}-cell bump channel ({ncell} cells per GPU, "
                              f"{a.nx}x{a.ny}x{a.nz} per GPU, wall-normal grading; " + ("linearised about the compressible primal CONVERGED on the GPU (Newton-Krylov, grid sequencing; subsonic perfect gas, p 101325 T 300 at the boundaries; |R| = %.2e)" % primal_residual_norm if (a.converge_primal and a.solver == "DARhoSimpleFoam") else "synthetic subsonic perfect-gas state p 101325 T 300") + (", one MRF zone with a rotating hub" if a.solver == "DATurboFoam" else "") + ")")
                             if a.solver != "DASimpleFoam" else
                             f"BASELINE configs[2]: DASimpleFoam+SA adjoint, {ncell}-cell hex mesh per GPU ({a.nx}x{a.ny}x{a.nz} bump channel, wall-normal "
                             f"grading; state: prolonged converged coarse primal)" if a.workload != "naca" else
                             f"BASELINE configs[2]: DASimpleFoam+SA adjoint, {n_global}-cell NACA0012 " + (f"SWEPT wing segment (sweep {a.naca_sweep:g} deg, taper {a.naca_taper:g}: every layer its own section; " if (a.naca_sweep or a.naca_taper) else "wing section (")
                             + f"O-grid{' numbered across the seam (renumbered like renumberMesh)' if a.naca_fold else ''}, {a.naca[0]} around x {a.naca[1]} normal x {a.naca[2]} "
                             f"spanwise hexahedra of {a.naca_dz} chords, first cell {a.naca_first_cell:g} chords, far field 20 chords, U 10 m/s, AoA 2 deg, Re 6.7e5; "
                             + ("synthetic noisy boundary-layer state)" if a.naca_synthetic else
                                "linearised about the primal CONVERGED on the GPU: Newton-Krylov, grid sequencing, |R| = %.2e)" % primal_residual_norm))
                            + f", full GMRES adjoint, {8 if a.solver == 'DASimpleFoam' else 9} states/cell, reference stencil tables; "
                            f"timed iterations sit at Krylov basis sizes j in [{j0}, {j0 + a.steps})"
                            + (" = around the mean basis depth of the full solve" if (mean_depth is not None and not a.window_at_warmup) else ""),
                "value_is": ("iterations/s of the K timed iterations placed at the mean basis depth of the full solve to 1e-6 (the orthogonalisation cost is linear in the depth, "
                             "so this is the mean rate of the whole solve; compare solve.iterations_per_sec_whole_solve)") if (mean_depth is not None and not a.window_at_warmup)
                            else "iterations/s of the K timed iterations at basis sizes [warmup, warmup + K)",
                "window_start_depth": j0,
                "primal_residual_norm": primal_residual_norm,
                "cells_per_gpu": ncell,
                "global_cells": n_global,
                "partition": (None if world == 1 else ("RCB cell partition of the global channel, 3 ghost rings (ShardedAdjointGeneral.scattered from rank 0)" if a.solver != "DASimpleFoam" else
                                                       {"span": "spanwise slabs of whole cell layers", "around": "sectors around the airfoil (whole spanwise columns per rank)",
                                                        "columns": "blocks of the (around, wall-normal) index plane (whole spanwise columns per rank)"}[a.naca_partition] + ", 3 ghost rings per cut (ShardedAdjointGeneral.scattered from rank 0)" if a.workload == "naca"
                                                       else "slabs along x, 3 ghost layers per cut (ShardedAdjoint)")),
                "global_solve_iterations_per_sec": a.steps * 1.0 / dt,
                "cell_iterations_per_sec": n_global * a.steps * 1.0 / dt,
                "aggregation": "value = iterations/s of the one global solve (weak scaling: flat = ideal); cell_iterations_per_sec = global cells x value",
                "states_per_gpu": n,
                "dRdWT_nnz": op_nnz,
                "dRdWT_structural_nnz": opmat_nnz,
                "colors": int(ncolors),
                "gmres_restart_window": window_restart,
                "cgs_refinements_in_window_run": int(L.das_ksp_get_n_refine(ksp.handle)) if hasattr(L, "das_ksp_get_n_refine") else None,
                "pc": pc_desc,
                "pc_factor_entries": fac_entries,
                "pc_coarse_aggregates": int(L.das_ksp_get_coarse(ksp.handle, None)),
                "pc_coarse_aggregates_global": int(global_coarse) if world > 1 else None,
                "pc_coarse_mode": D.getOption("amd")["pcCoarseMode"],
                "asm_overlap": (int(getattr(sharded, "asm_overlap", 0)) if sharded is not None else None),
                "pc_stability": pc_stab,
                "pc_upwind_blend": D.getOption("amd")["pcUpwindBlend"],
                "pc_options_passed_by_bench": sorted(k for k in make_opts(a, dev_index, 1, 1, 1e-6)["amd"] if k != "maxKrylovBytes" and not k.startswith("primal")),
                "coarse_ms": L.das_timer_avg_ms(h, b"coarse"),
                "halo_ms": L.das_timer_avg_ms(h, b"halo") if world > 1 else None,
                # adjoint setup (what the reference does between the primal and the Krylov solve) vs. building the synthetic input
                "setup_seconds": {"total": t_color + t_pcmat + t_pc + t_op, "coloring_and_connectivity": t_color, "dRdWTPC_fd_gpu": t_pcmat,
                                  "pc_factorisation_and_coarse_space": t_pc, "dRdWT_dual_gpu": t_op,
                                  "synthetic_mesh_and_state_generation_python": t_case},
                "dRdWTPsi_GBps": achieved,
                "dRdWTPsi_GBps_of_format_bytes": fmt_GBps,
                "spmv_ms": spmv_ms,
                "pc_apply_ms": pc_ms,
                "window_rel_residual": win_info["res"] / win_info["res0"] if win_info["res0"] else None,
                "solve": solve,
                "primal_newton_krylov": primal,
                "psi_parity_200k": parity,
            },
            "roofline": {
                "kernel": "k_spmv_vec3 + k_spmv_wave (dRdW^T.psi: U rows as packed group rows, scalar rows as CSR; fp64 values, int32 columns)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": pmc_traffic(op_nnz, "spmv"),  # separate rocprofv3 --pmc passes, committed under profiles/
                "launches_timed": int(spmv_cnt),
                "algorithmic_bytes_per_launch": spmv_bytes,
                "algorithmic_formula": "12 nnz + 4 (n+1) + 16 n (SURVEY.md 8d: CSR fp64 + int32)",
                "format_bytes_per_launch": fmt_bytes,
                "frac_of_format_bytes": (fmt_GBps / HBM_PEAK_GBS) if fmt_GBps else None,
            },
            "roofline_pc": {
                "kernel": "k_bilu_sweep x2 (forward + backward node-block sweeps)" if a.pctype == "bilu" else "k_ras_apply",
                "bound": "hbm",
                "achieved": pc_bytes / (pc_ms * 1e-3) / 1e9 if pc_ms and pc_ms > 0 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": pc_bytes / (pc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pc_ms and pc_ms > 0 else None,
                "algorithmic_bytes_per_launch": pc_bytes,
                "survey_formula_bytes": pc_bytes_survey,
                "survey_formula": "12 nnz(PC matrix) + 16 n (SURVEY.md 8d with the ILU(0) pattern = the PC matrix pattern)",
                "frac_of_survey_formula": pc_bytes_survey / (pc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pc_ms and pc_ms > 0 else None,
                "pc_matrix_nnz": pc_nnz,
                "traffic": (pmc_traffic(op_nnz, "k_bilu_sweep_forward") or 0.0) + (pmc_traffic(op_nnz, "k_bilu_sweep_backward") or 0.0) or None,
            },
            "roofline_iteration": {
                "bound": "hbm",
                "algorithmic_bytes_per_step": moved_bytes,
                "operator_products_per_step": products,
                "formula": "products x B_spmv + B_pc + 2 b j n + 48 n at the mean j of the window, b = bytes per stored basis entry (8; the split basis reads 4 + 8 in its two passes): what this "
                           "implementation moves (delayed re-orthogonalisation = 2 basis reads per iteration; products = operator products per step)"
                           if orth == "dcgs2" else "B_spmv + B_pc + 32 j n + 48 n (CGS with refinement: 4 basis reads)",
                "achieved": moved_bytes / (ms_step * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": moved_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "orthogonalization": orth,
                "baseline_md_model_bytes_per_step": iter_bytes,
                "frac_of_baseline_md_model": iter_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
        if _OVERRUN:  # a CPU leg is still inside its C code: leave without waiting for it
            sys.stderr.flush()
            os._exit(0)
    if world > 1:
        dist.destroy_process_group()
    return out


def _cpu_threads():
    """Threads of the CPU legs = the CPUs the process may really use (affinity AND the container's CFS quota, oracle.linear.available_cpus).
    Measured (round 4, profiles/r05m_host_cpu.txt): the bench host shows 256 CPUs, the container's quota is 16; STREAM triad 488 GB/s
    with 16 threads, 33-39 GB/s with 192-256 (throttled), and the level-parallel sweeps - ~2000 barriers per application - turn
    every barrier into a scheduler time slice once more threads run than the quota pays for."""
    if os.environ.get("DAS_BENCH_CPU_THREADS"):
        return int(os.environ["DAS_BENCH_CPU_THREADS"])
    from oracle.linear import available_cpus

    return available_cpus()


def _with_deadline(fn, seconds, what):
    """Run a CPU leg in a helper thread and give up after `seconds`: the JSON line must come out whatever the host does.  The leg's
    C code cannot be interrupted - a leg that overran keeps its thread, and main() leaves through os._exit after printing."""
    import threading

    box = {}

    def run():
        try:
            box["out"] = fn()
        except Exception as e:  # noqa: BLE001 - a baseline leg must never break the line
            box["out"] = {"error": repr(e)[:300]}

    th = threading.Thread(target=run, daemon=True)
    t0 = time.time()
    th.start()
    th.join(seconds)
    if th.is_alive():
        _OVERRUN.append(what)
        return {"error": f"{what}: not finished after {time.time() - t0:.0f} s (deadline {seconds:.0f} s); last stage: {_CPU_STAGE[0]}"}
    return box.get("out")


_OVERRUN = []
_CPU_STAGE = ["-"]


def _export(L, fn, handle, n, nnz):
    rp, ci, v = np.empty(n + 1, np.int64), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    rc = fn(handle, rp.ctypes.data_as(C.POINTER(C.c_longlong)), ci.ctypes.data_as(C.POINTER(C.c_int)), v.ctypes.data_as(C.POINTER(C.c_double)))
    if rc < 0:
        raise RuntimeError(L.das_last_error().decode()[:200])
    return rp, ci, v


def _cpu_solver(L, h, ksp, pc, n, N, op_nnz, pc_nnz, threads):
    """The oracle's all-core port loaded with the operator and the PC matrix of the GPU run (copied back from the device): CSR SpMV,
    the node-block ILU(0) restated for the host on the SAME node structure (KSP.pcStructure) and PC matrix, the same aggregates."""
    from oracle import linear as OL

    t0 = time.perf_counter()

    def mark(what):
        _CPU_STAGE[0] = f"{what} (+{time.perf_counter() - t0:.1f} s)"
        stage("   cpu leg: " + _CPU_STAGE[0])

    K = OL.OmpKrylov(threads)
    mark("device -> host copy of the operator")
    A = _export(L, L.das_op_export, h, n, op_nnz)
    mark("first-touch copy of the operator")
    K.set_operator(A)
    del A
    mark("device -> host copy of the PC matrix")
    P = _export(L, L.das_mat_export, pc.handle, n, pc_nnz)
    S = ksp.pcStructure()
    mark("node-block ILU(0) on the host: scatter + factorisation")
    K.set_pc_bilu(P, S)
    mark("coarse operator")
    nagg, agg = ksp.coarse(N)
    if nagg > 0 and agg.min() >= 0:
        K.set_coarse(P, 3 * N, N, agg)
    else:
        nagg = 0
    del P
    return K, dict(prep_seconds=time.perf_counter() - t0, ilu_levels=int(S["lvlPtr"].size - 1), nodes=int(S["nodeUnk"].shape[0]), blocks=int(S["bcol"].size),
                   shifted_pivots=int(K.nshift), coarse_aggregates=int(nagg))


def cpu_port_at_bench_size(a, L, h, ksp, pc, n, N, op_nnz, pc_nnz, rhs_h, gpu_spmv_ms, gpu_pc_ms):
    """CPU restatement (kind "port": the oracle's OpenMP C kernels, oracle/csrc/oracle_krylov_omp.c - NOT DAFoam) AT THE BENCH SIZE on
    all host cores: the operator dRdW^T and the PC matrix dRdWTPC of this very run are copied back from the device; right-
    preconditioned GMRES(CGS2) with the row-chunked first-touch SpMV, the SAME preconditioner restated for the host (node-block
    ILU(0) on the library's node structure, level-parallel factorisation and sweeps; + the same pressure coarse space, additive),
    threaded multi-dot / multi-axpy.  A bounded sample of iterations (about --cpu-seconds) at basis sizes j < sample: no extrapolation."""
    need = 12.0 * (op_nnz + pc_nnz) * 2.2 + 9.0 * 64 * 30 * N
    try:
        with open("/proc/meminfo") as f:
            avail = [int(ln.split()[1]) * 1024.0 for ln in f if ln.startswith("MemAvailable")][0]
    except (OSError, IndexError, ValueError):
        avail = 0.0
    if avail < need:
        return {"skipped": f"host MemAvailable {avail / 1e9:.0f} GB < {need / 1e9:.0f} GB"}
    threads = _cpu_threads()
    K, prep = _cpu_solver(L, h, ksp, pc, n, N, op_nnz, pc_nnz, threads)
    _CPU_STAGE[0] = "host STREAM triad"
    stream = K.stream_GBps(1 << 28, 5)
    _CPU_STAGE[0] = "GMRES pilot (4 iterations)"
    _, pilot = K.gmres(rhs_h, restart=4, fixed_iters=4)
    stage(f"   cpu leg: pilot {pilot['seconds'] / 4 * 1e3:.0f} ms per iteration (spmv {pilot['seconds_spmv'] / 5 * 1e3:.0f}, pc {pilot['seconds_pc'] / 5 * 1e3:.0f})")
    _CPU_STAGE[0] = "GMRES sample"
    per_it = pilot["seconds"] / 4
    iters = int(max(8, min(300, a.cpu_seconds / max(per_it, 1e-6))))
    _, inf = K.gmres(rhs_h, restart=iters, fixed_iters=iters)
    spmv_ms = inf["seconds_spmv"] / (iters + 1) * 1e3
    pc_ms = inf["seconds_pc"] / (iters + 1) * 1e3
    bytes_ = 12.0 * op_nnz + 4.0 * (n + 1) + 16.0 * n
    return {
        "value": iters / inf["seconds"],
        "unit": "iter/s",
        "cores": K.threads,
        "kind": "port",
        "sample": f"{iters} GMRES iterations at basis sizes j < {iters} (the GPU window sits at the mean depth of its full solve, several hundred vectors: its orthogonalisation "
                  f"works on a ~10x deeper basis - the comparison flatters the CPU) of the SAME {N}-cell system at the bench size - operator ({op_nnz} nnz) and PC matrix ({pc_nnz} nnz) "
                  f"copied back from the device - with the oracle's OpenMP C port (gcc -O3 -march=x86-64-v3 -fopenmp, {K.threads} threads = every CPU the container may use "
                  f"(affinity and CFS quota; the host shows {os.cpu_count()}), first-touch placement): CSR SpMV, "
                  f"the node-block ILU(0) of the GPU path restated for the host ({prep['nodes']} nodes, {prep['blocks']} dense 8x8 blocks, {prep['ilu_levels']} levels, "
                  f"level-parallel) + pressure coarse space ({prep['coarse_aggregates']} aggregates, additive), CGS2 with threaded multi-dot / multi-axpy; {inf['seconds']:.1f} s "
                  f"timed, prep {prep['prep_seconds']:.1f} s (untimed: device-to-host copies, block scatter, factorisation)",
        "host_cpus": os.cpu_count(),
        "usable_cpus": _cpu_threads(),  # affinity AND the container's CFS quota: `cores` = the threads actually used = this number
        "host_stream_triad_GBps": stream,
        "seconds_timed": inf["seconds"],
        "iterations_timed": iters,
        "ms_per_iteration": inf["seconds"] / iters * 1e3,
        "dRdWTPsi_at_bench_size": {"ms": spmv_ms, "GBps": bytes_ / (spmv_ms * 1e-3) / 1e9 if spmv_ms > 0 else None, "gpu_ms": gpu_spmv_ms},
        "pc_apply_at_bench_size": {"ms": pc_ms, "gpu_ms": gpu_pc_ms, "levels": prep["ilu_levels"]},
        "orthogonalisation_ms_mean": inf["seconds_orth"] / iters * 1e3,
        "prep": prep,
    }


def psi_parity_200k(a, dev_index, case2d=None):
    """BASELINE.md section 3 / VERDICT round 3 item 2: the adjoint vector of a 200 k-cell system of the bench family (naca: the
    bench's converged section extruded to 16 spanwise layers = 198 k cells, BASELINE configs[1] size, polished on the extruded mesh;
    channel: 100 x 50 x 40) solved to --parity-tol by the GPU path and, independently, by the host port (OpenMP C: CSR SpMV, the
    node-block ILU(0) restated for the host, GMRES(CGS2)) on the exported matrices: |psi_gpu - psi_cpu| / |psi_cpu| (north_star bar
    1e-6; measured 1.0e-12 with both sides at 1e-10, tests/test_gpu_naca.py)."""
    from dafoam_amd import _capi
    from dafoam_amd.meshgen import bench_channel_case
    from dafoam_amd.pyDAFoam import PYDAFOAM
    from dafoam_amd.pyDASolvers import KSP, Mat, Vec

    L = _capi.lib()
    if a.workload == "naca":
        from dafoam_amd.workloads import naca_converged_primal, naca_extruded_case

        o10 = make_opts(a, dev_index, 1000, 1000, a.parity_tol)
        if case2d is None:
            case2d, _ = naca_converged_primal(a.naca[0], a.naca[1], options=o10, first_cell=a.naca_first_cell, case_kwargs=({"fold_seam": True} if a.naca_fold else None))
        nzp = max(1, int(round(200000.0 / (a.naca[0] * a.naca[1]))))
        # (the section arrives in the bench's cell numbering: the extrusion must use the same one)
        case, _ = (naca_extruded_case(case2d, (a.naca[0], a.naca[1]), nzp, dz=0.1, first_cell=a.naca_first_cell, options=o10,
                                      case_kwargs=({"fold_seam": True} if a.naca_fold else None)) if nzp > 1 else (case2d, None))
        what = f"NACA0012 wing section {a.naca[0]} x {a.naca[1]} x {nzp} (the bench's converged section, {nzp} spanwise layers of 0.1 chords: BASELINE configs[1] size)"
    else:
        case, what = bench_channel_case(100, 50, 40), "bump channel 100 x 50 x 40"
    D = PYDAFOAM(options=make_opts(a, dev_index, 2000, 2000, a.parity_tol), case=case)  # restart 2000: no restart inside the plateau of the residual history
    n, N = D.getNLocalAdjointStates(), case.mesh.n_cells
    tj = time.perf_counter()
    D.solver.runColoring()
    t_gcol = time.perf_counter() - tj
    tj = time.perf_counter()
    P = Mat()
    D.solver.calcdRdWT(1, P)
    t_gpc = time.perf_counter() - tj
    ksp = KSP()
    D.solverAD.createMLRKSPMatrixFree(P, ksp)
    tj = time.perf_counter()
    D.solverAD.initializedRdWTMatrixFree()
    t_gop = time.perf_counter() - tj
    rhs = np.zeros(n)
    rhs[0 : 3 * N : 3] = 1.0 / N
    b, x = Vec(n), Vec(n)
    b.array[:] = rhs
    t0 = time.perf_counter()
    gfail = D.solverAD.solveLinearEqn(ksp, b, x)
    t_gpu = time.perf_counter() - t0
    ginf = ksp.info()
    psi_gpu = x.array.copy()
    threads = _cpu_threads()
    # the CPU side ASSEMBLES ITS OWN matrices (oracle/adjoint_host.py via oracle/parity_host.py: connectivity from the stencil tables, own
    # colouring, the face-based residual port with dual numbers / finite differences) - no matrix of the device run is exported
    from oracle.parity_host import host_adjoint_solve

    def mark(what):
        _CPU_STAGE[0] = what
        stage("   parity leg: " + what)

    blend = float(D.getOption("amd").get("pcUpwindBlend", 0.0))
    psi_cpu, cinf = host_adjoint_solve(case, NORM, rhs, ksp.pcStructure(), ksp.coarse(N), threads, rel_tol=a.parity_tol, pc_blend=blend, restart=1500, max_iters=3000,
                                       max_seconds=float(os.environ.get("DAS_BENCH_PARITY_CPU_SECONDS", 200)), stage=mark)
    return {"system": what, "cells": int(N), "states": int(n), "rel_tol_both": a.parity_tol,
            "gpu": {"iterations": int(ginf["iters"]), "seconds": t_gpu, "fail": int(gfail), "rel_residual": ginf["res"] / ginf["res0"] if ginf["res0"] else None,
                    "jacobian_build_seconds": {"connectivity_and_colouring": t_gcol, "dRdWTPC_fd": t_gpc, "dRdWT_dual": t_gop, "total": t_gcol + t_gpc + t_gop}},
            "cpu": {"matrices": "host-assembled", "iterations": int(cinf["iters"]), "seconds": cinf["seconds"], "fail": int(cinf["fail"]),
                    "rel_residual": cinf["res"] / cinf["res0"] if cinf["res0"] else None, "threads": cinf["threads"], "gmresRestart": 1500,
                    "colors": cinf["colors"], "dRdWT_nnz": cinf["dRdWT_nnz"], "dRdWTPC_nnz": cinf["dRdWTPC_nnz"],
                    "jacobian_build_seconds": {"connectivity_and_colouring": cinf["connectivity_and_colouring_s"], "dRdWTPC_fd": cinf["dRdWTPC_fd_s"],
                                               "dRdWT_dual": cinf["dRdWT_dual_s"], "total": cinf["jacobian_build_s"]},
                    "factorisation_seconds": cinf["factorisation_s"],
                    "assembled_by": "oracle/adjoint_host.py (OpenMP C++: stencil-table connectivity, first-fit colouring, face-based DASimpleFoam+SA residual, dual numbers for "
                                    "dRdW^T, one-sided differences 1e-6 for dRdWTPC); solved by oracle/csrc/oracle_krylov_omp.c",
                    "pc": "node-block ILU(0) of the HOST-assembled dRdWTPC on the library's node structure (integer tables) + additive pressure coarse space"},
            "psi_rel_diff_gpu_vs_cpu": float(np.linalg.norm(psi_gpu - psi_cpu) / np.linalg.norm(psi_cpu)), "bar": 1e-6}


if __name__ == "__main__":
    main()
