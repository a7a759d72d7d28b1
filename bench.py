#!/usr/bin/env python
"""bench.py - adjoint hot-path benchmark (BASELINE.json metric: adjoint GMRES iterations/s + dRdWTPsi GB/s).

Workload at N = 1: BASELINE configs[2] - DASimpleFoam + SA, 2 M-cell hex mesh (250x100x80 bump channel until the
swept-wing generator exists), one GPU, full GMRES adjoint.  One "step" = one right-preconditioned GMRES iteration of
the adjoint solve: node-block ILU(0) apply (two sync-free triangular sweeps) + dRdW^T.z SpMV + CGS (refine-if-needed)
orthogonalisation against the j basis vectors + norm, on the device-resident system assembled by coloured dual-number /
FD perturbation of the HIP residual.  Matrices, rhs and Krylov basis are resident in HBM when the timed region starts.

Timed region: the solve is advanced W (= --warmup) iterations inside ONE Arnoldi cycle, then EXACTLY K (= --steps)
iterations are timed, i.e. at basis sizes j in [W, W+K) (defaults 100 and 100: the orthogonalisation cost of a realistic
solve, not of its first iterations).  Afterwards (N = 1) the same system is solved from scratch to gmresRelTol = 1e-6
with the reference's defaults (gmresRestart = gmresMaxIters = 1000): `config.solve` reports
iterations, time_to_tolerance_s and the reference's fail flag (DALinearEqn.C:422-434).

  python bench.py --gpus 1 --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0) with `roofline` (dRdW^T.psi SpMV, HIP-event timed on the launch stream), `roofline_pc`,
`roofline_iteration` and `cpu_baseline` (the oracle's C kernels on the host cores, bounded sample).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--nx", type=int, default=int(os.environ.get("DAS_BENCH_NX", 250)))
    ap.add_argument("--ny", type=int, default=int(os.environ.get("DAS_BENCH_NY", 100)))
    ap.add_argument("--nz", type=int, default=int(os.environ.get("DAS_BENCH_NZ", 80)))
    ap.add_argument("--workload", default=os.environ.get("DAS_BENCH_WORKLOAD", "channel"),
                    help="channel: nx x ny x nz bump channel (default, state prolonged from a converged coarse primal); naca: NACA0012 O-grid of "
                         "--naca n_around n_normal nz cells extruded in span (synthetic boundary-layer state), N = 1 only")
    ap.add_argument("--naca", type=int, nargs=3, default=[800, 250, 10])
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("DAS_BENCH_CPU_SECONDS", 15.0)))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the solve-to-tolerance phase")
    ap.add_argument("--pctype", default=os.environ.get("DAS_BENCH_PCTYPE", "bilu"))
    ap.add_argument("--fp32-factor", type=int, default=int(os.environ.get("DAS_BENCH_PCFP32", 0)))
    ap.add_argument("--krylov-gb", type=float, default=float(os.environ.get("DAS_BENCH_KRYLOV_GB", 160.0)))
    ap.add_argument("--solve-restart", type=int, default=1000)
    ap.add_argument("--solve-maxit", type=int, default=1000)
    ap.add_argument("--converge-primal", action="store_true", help="converge the flow state with the GPU Newton-Krylov primal before the adjoint (opt-in: the adjoint's conditioning does not depend on it, DESIGN.md section 6b)")
    ap.add_argument("--cpu-solve", action="store_true", help="cpu_baseline additionally solves the 200 k-cell sample to 1e-6 on the host cores and reports time-to-tolerance and |psi_gpu - psi_cpu| (minutes)")
    ap.add_argument("--coarse-agg", type=int, default=int(os.environ.get("DAS_BENCH_COARSE", -1)), help="two-level PC: aggregates (-1 auto, 0 off)")
    ap.add_argument("--coarse-mode", default=os.environ.get("DAS_BENCH_COARSE_MODE", "additive"))
    ap.add_argument("--orth", default=os.environ.get("DAS_BENCH_ORTH", "dcgs2"), help="dcgs2 (delayed re-orthogonalisation, 2 basis reads / iteration) | cgs (reference: refine if needed)")
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


def make_opts(a, dev_index, restart, maxit, rtol):
    return {
        "solverName": "DASimpleFoam",
        "normalizeStates": dict(NORM),
        "adjEqnOption": {"gmresRestart": int(restart), "gmresMaxIters": int(maxit), "gmresRelTol": rtol, "gmresAbsTol": 1e-300, "printInfo": 0},
        "amd": {"pcType": a.pctype, "pcFactorFP32": a.fp32_factor, "maxKrylovBytes": int(a.krylov_gb * 2**30),
                "pcCoarseAggregates": a.coarse_agg, "pcCoarseMode": a.coarse_mode, "gmresOrthogonalization": a.orth},
        "amdDevice": dev_index,
    }


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # debugging aids for boxes with a single GPU (never set by the driver): all ranks on device 0, host-staged gloo
    backend = os.environ.get("DAS_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DAS_BENCH_ONE_GPU") == "1" else local_rank
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
    window_restart = max(a.steps + a.warmup, 1)
    opts = make_opts(a, dev_index, window_restart, 10**9, 1e-30)
    sharded = None
    if world > 1:
        # weak scaling: the global channel has nx*world cell columns, every rank owns nx of them (+3 ghost layers)
        from dafoam_amd.distributed import ShardedAdjoint

        sharded = ShardedAdjoint(a.nx * world, a.ny, a.nz, opts, device_index=dev_index)
        D = sharded.D
        case = sharded.case
        ncell = a.nx * a.ny * a.nz
    else:
        if a.workload == "naca":
            from dafoam_amd.meshgen import naca0012_case

            case = naca0012_case(a.naca[0], a.naca[1], a.naca[2], span=0.1 * a.naca[2])
        else:
            # state = prolongation of a converged coarse primal (dafoam_amd/data/channel_primal_coarse.npz)
            case = bench_channel_case(a.nx, a.ny, a.nz)
        ncell = case.mesh.n_cells
        D = PYDAFOAM(options=opts, case=case)
    t_case = time.time() - t_setup
    primal = None
    if a.converge_primal and world == 1:
        t0 = time.time()
        D.setOption("primalMinResTol", 1e-8) if hasattr(D, "setOption") else None
        pf = D.solvePrimal(maxSteps=100)
        primal = dict(D.primalInfo, fail=int(pf), seconds=time.time() - t0)
        primal["history"] = [float(v) for v in primal["history"]]
    h = D.solver._h
    n = D.getNLocalAdjointStates()
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
    rhs_h[0 : 3 * N : 3] = 1.0 / (N * world)
    if sharded is not None:
        rhs_h = np.where(sharded.owned, rhs_h, 0.0)
    rhs = torch.from_numpy(rhs_h).cuda()
    sol = torch.zeros(n, dtype=torch.float64, device="cuda")
    setup_s = time.time() - t_setup

    def check(rc):
        if rc < 0:
            raise RuntimeError(L.das_last_error().decode())
        return rc

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed window: iterations j in [W, W+K) of one Arnoldi cycle -------------------------------------------------
    check(L.das_ksp_begin_device(h, ksp.handle, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()), 1))
    if a.warmup > 0:
        check(L.das_ksp_advance(h, ksp.handle, int(a.warmup)))
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
    if a.pctype == "bilu":
        # dense 8x8 node blocks: 8 (4) B per factor entry, one int32 per block, + b, y, z, out vectors
        pc_bytes = (4.0 if a.fp32_factor else 8.0) * fac_entries + 4.0 * fac_entries / 64.0 + 8.0 * (2 * n + 3 * n_ext)
    else:
        pc_bytes = 12.0 * fac_entries + 16.0 * n_ext
    jmean = a.warmup + 0.5 * a.steps
    iter_bytes = spmv_bytes + pc_bytes + 32.0 * jmean * n + 48.0 * n  # BASELINE.md section 3 (CGS with refinement: 4 basis reads)
    # what this implementation has to move: the delayed re-orthogonalisation reads the basis twice per iteration
    orth = a.orth
    moved_bytes = spmv_bytes + pc_bytes + (16.0 if orth == "dcgs2" else 32.0) * jmean * n + 48.0 * n
    ms_step = dt / a.steps * 1e3

    # ---- solve to tolerance (N = 1): the reference's defaults ----------------------------------------------------------
    solve = None
    if not a.no_solve:  # N > 1: the same calls on every rank - ONE global solve (collective inside the library)
        # the reference's defaults: gmresRestart = gmresMaxIters = 1000 (pyDAFoam.py:526-548)
        D.solver.updateDAOption({"adjEqnOption": {"gmresRestart": a.solve_restart, "gmresMaxIters": a.solve_maxit, "gmresRelTol": 1e-6, "gmresAbsTol": 1e-14}})
        sol.zero_()
        barrier()
        t0 = time.perf_counter()
        check(L.das_ksp_begin_device(h, ksp.handle, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()), 0))
        while not check(L.das_ksp_advance(h, ksp.handle, 1000)):
            pass
        fail = check(L.das_ksp_end(h, ksp.handle))
        barrier()
        t_solve = time.perf_counter() - t0
        inf = ksp.info()
        hist = ksp.history()
        solve = {"converged": fail == 0, "fail": int(fail), "iterations": inf["iters"], "time_to_tolerance_s": t_solve,
                 "rel_residual": inf["res"] / inf["res0"] if inf["res0"] else None, "gmresRelTol": 1e-6,
                 "gmresRestart": int(min(a.solve_restart, a.krylov_gb * 2**30 // (8 * n) - 1)), "gmresMaxIters": a.solve_maxit,
                 "rel_residual_at_1000_iterations": float(hist[1000] / hist[0]) if len(hist) > 1000 else None,
                 "rel_residual_every_100": [float(v / hist[0]) for v in hist[::100]],
                 "iterations_per_sec_whole_solve": inf["iters"] / t_solve}

    out = None
    if rank == 0:
        cpu = None
        if not a.no_cpu and world == 1:
            try:
                cpu = cpu_baseline(a, dev_index, ncell)
            except Exception as e:  # noqa: BLE001 - the baseline must never break the line
                cpu = {"error": str(e)[:300]}
            try:  # per-component figure AT the bench size (no "scaled by 0.1" for the product itself)
                cpu["dRdWTPsi_at_bench_size"] = cpu_spmv_at_bench_size(L, h, n, op_nnz)
                if "ms" in cpu["dRdWTPsi_at_bench_size"] and spmv_ms:
                    cpu["dRdWTPsi_at_bench_size"]["gpu_ms"] = spmv_ms
            except Exception as e:  # noqa: BLE001
                cpu["dRdWTPsi_at_bench_size"] = {"error": str(e)[:300]}
        pc_desc = ("node-block ILU(0) of FD dRdWTPC over the whole rank (8-slot cell nodes, 8x8 fp%s blocks), factorised on the device, "
                   "two sync-free sweeps per apply; + piecewise-constant pressure coarse space (%s)" % ("32" if a.fp32_factor else "64", a.coarse_mode)) if a.pctype == "bilu" else \
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"BASELINE configs[2]: DASimpleFoam+SA adjoint, {ncell}-cell hex mesh per GPU ({a.nx}x{a.ny}x{a.nz} bump channel, wall-normal "
                             f"grading; state: prolonged converged coarse primal)" if a.workload != "naca" else
                             f"BASELINE configs[2]: DASimpleFoam+SA adjoint, {ncell}-cell NACA0012 O-grid ({a.naca[0]} around x {a.naca[1]} normal x {a.naca[2]} "
                             f"spanwise hexahedra, first cell 2e-5 chords, far field 20 chords; synthetic boundary-layer state)")
                            + f", full GMRES adjoint, 8 states/cell, reference stencil tables; "
                            f"timed iterations sit at Krylov basis sizes j in [{a.warmup}, {a.warmup + a.steps})",
                "cells_per_gpu": ncell,
                "global_cells": ncell * world,
                "global_solve_iterations_per_sec": a.steps * 1.0 / dt,
                "cell_iterations_per_sec": ncell * world * a.steps * 1.0 / dt,
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
                "pc_coarse_mode": a.coarse_mode,
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
                "traffic": (pmc_traffic(op_nnz, "k_bilu_sweep_forward") or 0.0) + (pmc_traffic(op_nnz, "k_bilu_sweep_backward") or 0.0) or None,
            },
            "roofline_iteration": {
                "bound": "hbm",
                "algorithmic_bytes_per_step": moved_bytes,
                "formula": "B_spmv + B_pc + 16 j n + 48 n at the mean j of the window: what this implementation moves (delayed re-orthogonalisation = 2 basis reads per iteration)"
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
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return out


def cpu_spmv_at_bench_size(L, h, n, op_nnz, seconds=8.0):
    """dRdW^T.psi of the SAME operator on the host cores, at the bench size (no extrapolation): the CSR arrays are copied back
    from the device (das_op_export) and multiplied by the oracle's C kernel, one contiguous row chunk of equal nnz per thread
    (ctypes releases the GIL).  Skipped when the host cannot hold the matrix twice over."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import linear as OL

    need = 12.0 * op_nnz + 8.0 * n
    try:
        with open("/proc/meminfo") as f:
            avail = [int(ln.split()[1]) * 1024.0 for ln in f if ln.startswith("MemAvailable")][0]
    except (OSError, IndexError, ValueError):
        avail = 0.0
    if avail < 2.5 * need:
        return {"skipped": f"host MemAvailable {avail / 1e9:.0f} GB < 2.5 x {need / 1e9:.0f} GB"}
    t0 = time.perf_counter()
    rp, ci, v = np.empty(n + 1, np.int64), np.empty(op_nnz, np.int32), np.empty(op_nnz, np.float64)
    rc = L.das_op_export(h, rp.ctypes.data_as(C.POINTER(C.c_longlong)), ci.ctypes.data_as(C.POINTER(C.c_int)), v.ctypes.data_as(C.POINTER(C.c_double)))
    if rc < 0:
        return {"skipped": L.das_last_error().decode()[:200]}
    t_copy = time.perf_counter() - t0
    threads = max(1, min(64, os.cpu_count() or 1))
    cuts = np.searchsorted(rp, np.linspace(0, op_nnz, threads + 1))
    cuts[0], cuts[-1] = 0, n
    lib = OL.lib()
    x = np.random.default_rng(2).uniform(-1.0, 1.0, n)
    y = np.empty(n)
    lp, ip, dp = C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_double)
    xp, cip, vp = x.ctypes.data_as(dp), ci.ctypes.data_as(ip), v.ctypes.data_as(dp)

    def chunk(t):
        a, b = int(cuts[t]), int(cuts[t + 1])
        if b > a:  # rp holds absolute offsets: a row range needs no copy
            lib.csr_spmv(b - a, C.cast(C.addressof(rp.ctypes.data_as(lp).contents) + 8 * a, lp), cip, vp, xp,
                         C.cast(C.addressof(y.ctypes.data_as(dp).contents) + 8 * a, dp))

    pool = ThreadPoolExecutor(max_workers=threads)
    list(pool.map(chunk, range(threads)))  # warm-up (page faults of y)
    reps, t1 = 0, time.perf_counter()
    while reps < 3 or time.perf_counter() - t1 < seconds:
        list(pool.map(chunk, range(threads)))
        reps += 1
        if reps >= 50:
            break
    dt = (time.perf_counter() - t1) / reps
    pool.shutdown()
    bytes_ = 12.0 * op_nnz + 4.0 * (n + 1) + 16.0 * n
    return {"ms": dt * 1e3, "GBps": bytes_ / dt / 1e9, "threads": threads, "repetitions": reps, "device_to_host_copy_s": t_copy,
            "what": "the bench operator itself (CSR copied back from the device), oracle C kernel csr_spmv, one row chunk of equal nnz per thread"}


def cpu_baseline(a, dev_index, ncell_gpu):
    """CPU restatement (kind "port": the oracle's C kernels, NOT DAFoam) on the host cores, bounded sample: the same solver
    on a 10x smaller mesh of the same family (100x50x40 = 200 k cells) - its matrices are assembled by the GPU path and
    copied back - runs right-preconditioned GMRES with a row-chunked SpMV and block-Jacobi ILU(1) (one block per thread:
    the reference's one-ASM-sub-domain-per-MPI-rank layout) for about `--cpu-seconds`; iterations/s is scaled to the
    GPU mesh by the cell ratio (every per-iteration cost of this algorithm is linear in the cell count)."""
    from oracle import linear as OL
    from dafoam_amd.meshgen import bench_channel_case
    from dafoam_amd.pyDAFoam import PYDAFOAM
    from dafoam_amd.pyDASolvers import Mat

    t0 = time.time()
    dims = (100, 50, 40)
    case = bench_channel_case(*dims)
    D = PYDAFOAM(options=make_opts(a, dev_index, 50, 50, 1e-30), case=case)
    D.solver.runColoring()
    P = Mat()
    D.solver.calcdRdWT(1, P)
    A = Mat()
    D.solver.calcdRdWT(0, A, mode=1)
    Ah, Ph = A.to_scipy(), P.to_scipy()
    A.destroy()
    P.destroy()
    n = Ah.shape[0]
    N = case.mesh.n_cells
    rhs = np.zeros(n)
    rhs[0 : 3 * N : 3] = 1.0 / N
    threads = max(1, min(64, (os.cpu_count() or 1)))
    T = OL.ThreadedOperators(Ah, Ph, threads, fill=1)
    prep = time.time() - t0
    t1 = time.perf_counter()
    T.matvec(rhs)
    t_spmv = time.perf_counter() - t1
    # pilot of 3 iterations sizes the bounded sample
    t1 = time.perf_counter()
    OL.gmres(T.matvec, rhs, T.pc_solve, restart=3, fixed_iters=3)
    per_it = (time.perf_counter() - t1) / 3
    iters = int(max(5, min(200, a.cpu_seconds / max(per_it, 1e-6))))
    t1 = time.perf_counter()
    OL.gmres(T.matvec, rhs, T.pc_solve, restart=iters, fixed_iters=iters)
    dt = time.perf_counter() - t1
    ratio = N / float(ncell_gpu)
    extra = {}
    if a.cpu_solve:
        # time-to-tolerance of the CPU restatement and ||psi_GPU - psi_CPU|| / ||psi_CPU|| on the same
        # system, both solved to 1e-10 (a 1e-6 stop would leave a difference bounded by the conditioning, not by parity);
        # 50x25x20 = 25 k cells keeps the serial Gram-Schmidt of the CPU port within a minute
        dims2 = (50, 25, 20)
        case2 = bench_channel_case(*dims2)
        D2 = PYDAFOAM(options=make_opts(a, dev_index, 1000, 1000, 1e-10), case=case2)
        D2.solver.runColoring()
        P2, A2 = Mat(), Mat()
        D2.solver.calcdRdWT(1, P2)
        D2.solver.calcdRdWT(0, A2, mode=1)
        Ah2, Ph2 = A2.to_scipy(), P2.to_scipy()
        N2 = case2.mesh.n_cells
        rhs2 = np.zeros(Ah2.shape[0])
        rhs2[0 : 3 * N2 : 3] = 1.0 / N2
        T2 = OL.ThreadedOperators(Ah2, Ph2, threads, fill=1)
        t1 = time.perf_counter()
        psi_cpu, info = OL.gmres(T2.matvec, rhs2, T2.pc_solve, restart=1000, max_iters=1000, rel_tol=1e-10, abs_tol=1e-300)
        t_cpu = time.perf_counter() - t1
        D2.solver.updateDAOption({"adjEqnOption": {"gmresAbsTol": 1e-300}})
        t1 = time.perf_counter()
        psi_gpu, gfail = D2.solveAdjoint(rhs2)
        t_gpu = time.perf_counter() - t1
        extra = {"solve_to_1e-10_25k_cells": {"cpu_iterations": int(info["iters"]), "cpu_seconds": t_cpu, "cpu_rel_residual": float(info["res"] / info["res0"]),
                                              "gpu_iterations": int(D2.ksp.info()["iters"]), "gpu_seconds_incl_setup": t_gpu, "gpu_fail": int(gfail),
                                              "psi_rel_diff_gpu_vs_cpu": float(np.linalg.norm(psi_gpu - psi_cpu) / np.linalg.norm(psi_cpu))}}
    return {**extra, 
        "value": iters / dt * ratio,
        "unit": "iter/s",
        "cores": T.threads,
        "kind": "port",
        "sample": f"{iters} GMRES iterations at basis sizes j < {iters} on a {dims[0]}x{dims[1]}x{dims[2]} = {N}-cell mesh of the same family (matrices assembled "
                  f"on the GPU, copied back): oracle C kernels (gcc -O3 -march=native), {T.threads} threads, row-chunked SpMV + block-Jacobi ILU(1) "
                  f"(one block per thread), serial CGS2; measured {iters / dt:.2f} iter/s at {N} cells, scaled by {ratio:.3f} to the {ncell_gpu}-cell GPU workload; "
                  f"one SpMV {t_spmv * 1e3:.1f} ms; prep {prep:.1f} s (untimed)",
        "measured_iter_per_sec_at_sample_size": iters / dt,
        "sample_cells": N,
        "spmv_GBps": (12.0 * Ah.nnz + 4.0 * (n + 1) + 16.0 * n) / t_spmv / 1e9,
        "host_cpus": os.cpu_count(),
    }


if __name__ == "__main__":
    main()
