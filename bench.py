#!/usr/bin/env python
"""bench.py - adjoint hot-path benchmark (BASELINE.json metric: adjoint GMRES iterations/s + dRdWTPsi GB/s).

One "step" = one right-preconditioned GMRES iteration of the adjoint solve (block-ILU(0) apply + dRdW^T.z SpMV +
CGS2 orthogonalisation + norm) on the device-resident system assembled by coloured dual-number / FD perturbation
of the HIP residual.  Inputs (matrices, rhs, Krylov basis) are resident in HBM when the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = transposed-CSR SpMV, HIP-event timed on the
launch stream) and `cpu_baseline` (the oracle's C kernels on the host, bounded sample).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nx", type=int, default=int(os.environ.get("DAS_BENCH_NX", 100)))
    ap.add_argument("--ny", type=int, default=int(os.environ.get("DAS_BENCH_NY", 50)))
    ap.add_argument("--nz", type=int, default=int(os.environ.get("DAS_BENCH_NZ", 40)))
    ap.add_argument("--cpu-sample-iters", type=int, default=int(os.environ.get("DAS_BENCH_CPU_ITERS", 6)))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--block", type=int, default=int(os.environ.get("DAS_BENCH_BLOCK", 1024)))
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("DAS_BENCH_OVERLAP", 1)))
    ap.add_argument("--fill", type=int, default=int(os.environ.get("DAS_BENCH_FILL", 1)))
    return ap.parse_args()


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

    # host-side setup (pattern build, ILU factorisation) is OpenMP-parallel: give every rank its share of the cores
    # (torch.distributed.run exports OMP_NUM_THREADS=1 when it is unset; the library reads it when it is loaded below)
    if world > 1:
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

    t_setup = time.time()
    opts = {
        "solverName": "DASimpleFoam",
        "normalizeStates": {"U": 10.0, "p": 50.0, "nuTilda": 1e-3, "phi": 1.0},
        "adjEqnOption": {"gmresRestart": max(a.steps, a.warmup, 1), "gmresMaxIters": 100000, "gmresRelTol": 1e-30,
                         "gmresAbsTol": 1e-300, "printInfo": 0, "asmOverlap": a.overlap, "pcFillLevel": a.fill},
        "amd": {"pcBlockCells": a.block},
        "amdDevice": dev_index,
    }
    L = _capi.lib()
    sharded = None
    if world > 1:
        # weak scaling: the global channel has nx*world cell columns, every rank owns nx of them (+3 ghost layers);
        # halo reduction over RCCL p2p, dots over RCCL all-reduce (dafoam_amd/distributed.py)
        from dafoam_amd.distributed import ShardedAdjoint

        sharded = ShardedAdjoint(a.nx * world, a.ny, a.nz, opts, device_index=dev_index)
        D = sharded.D
        case = sharded.case
        ncell = a.nx * a.ny * a.nz
    else:
        # state = prolongation of a converged coarse primal (dafoam_amd/data/channel_primal_coarse.npz)
        case = bench_channel_case(a.nx, a.ny, a.nz)
        ncell = case.mesh.n_cells
        D = PYDAFOAM(options=opts, case=case)
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
    t_ilu = time.time() - t0
    t0 = time.time()
    D.solverAD.initializedRdWTMatrixFree()
    t_op = time.time() - t0
    # rhs on the device (torch owns the buffers; the C-ABI gets raw pointers)
    rng = np.random.default_rng(1234 + rank)
    rhs_h = rng.standard_normal(n)
    if sharded is not None:
        rhs_h = np.where(sharded.owned, rhs_h, 0.0)
    rhs = torch.from_numpy(rhs_h).cuda()
    sol = torch.zeros(n, dtype=torch.float64, device="cuda")
    setup_s = time.time() - t_setup

    def run(iters):
        rc = L.das_ksp_run_fixed_device(h, ksp.handle, C.c_void_p(rhs.data_ptr()), C.c_void_p(sol.data_ptr()), int(iters))
        if rc < 0:
            raise RuntimeError(L.das_last_error().decode())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup > 0:
        run(a.warmup)
    L.das_timer_reset(h)
    L.das_timer_enable(h, 1)
    barrier()
    t0 = time.perf_counter()
    run(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    L.das_timer_enable(h, 0)
    spmv_ms = L.das_timer_avg_ms(h, b"spmv")
    spmv_cnt = L.das_timer_count(h, b"spmv")
    pc_ms = L.das_timer_avg_ms(h, b"pc")
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # operator size (for the algorithmic-bytes formula of SURVEY.md section 8d / BASELINE.md section 3)
    opmat_nnz = int(L.das_get_con_nnz(h, 0))
    # the operator drops exact zeros (jacLowerBounds 1e-30): use its true nnz
    op_nnz = int(L.das_op_nnz(h))
    spmv_bytes = 12.0 * op_nnz + 4.0 * (n + 1) + 16.0 * n
    achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9 if spmv_ms and spmv_ms > 0 else None
    pc_bytes = 12.0 * L.das_ksp_get_factor_nnz(ksp.handle) + 16.0 * L.das_ksp_get_n_ext(ksp.handle)

    out = None
    if rank == 0:
        cpu = None
        if not a.no_cpu and world == 1:
            cpu = cpu_baseline(D, pc, a.cpu_sample_iters, n)
        traffic = None
        tf = os.path.join(ROOT, "profiles", "spmv_traffic_bytes.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "adjoint_gmres_iterations_per_sec",
            # whole-job aggregate: every rank advances its 200k-cell shard through `steps` GMRES iterations of ONE global
            # solve (weak scaling: N x more cells per iteration), so the job processes world * steps shard-iterations; at
            # N = 1 this is the plain iterations/s of the solve.  config.global_solve_iterations_per_sec is steps / time.
            "value": world * a.steps * 1.0 / dt,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"DASimpleFoam+SA adjoint, bump-channel hex mesh (state: prolonged converged coarse primal) {a.nx}x{a.ny}x{a.nz} = {ncell} cells per GPU "
                            f"(stand-in for BASELINE configs[1] NACA0012 ~200k cells: same solver, 8 states/cell, reference stencil tables)",
                "cells_per_gpu": ncell,
                "global_cells": ncell * world,
                "global_solve_iterations_per_sec": a.steps * 1.0 / dt,
                "aggregation": "value = n_gpus x global_solve_iterations_per_sec: GMRES iterations/s per 200k-cell shard summed over the shards of one global solve",
                "states_per_gpu": n,
                "dRdWT_nnz": op_nnz,
                "dRdWT_structural_nnz": opmat_nnz,
                "colors": int(ncolors),
                "gmres_restart": max(a.steps, a.warmup, 1),
                "pc": f"RAS(overlap {a.overlap} cell ring)+ILU({a.fill}) of FD dRdWTPC, RCB blocks of <= {a.block} cells, one workgroup per block",
                "halo_ms": L.das_timer_avg_ms(h, b"halo") if world > 1 else None,
                "setup_seconds": {"total": setup_s, "coloring_host": t_color, "dRdWTPC_fd": t_pcmat, "ilu_host": t_ilu, "dRdWT_dual": t_op},
                "dRdWTPsi_GBps": achieved,
                "spmv_ms": spmv_ms,
                "pc_apply_ms": pc_ms,
            },
            "roofline": {
                "kernel": "k_spmv (dRdW^T.psi, transposed CSR fp64/int32)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic,
                "launches_timed": int(spmv_cnt),
                "algorithmic_bytes_per_launch": spmv_bytes,
            },
            "roofline_pc": {
                "kernel": "k_ras_apply (RAS+ILU(k) level-scheduled triangular solves; largest share of an iteration)",
                "bound": "hbm",
                "achieved": pc_bytes / (pc_ms * 1e-3) / 1e9 if pc_ms and pc_ms > 0 else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": pc_bytes / (pc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pc_ms and pc_ms > 0 else None,
                "algorithmic_bytes_per_launch": pc_bytes,
                "note": "12 B per factor entry (fp64 value + 2x u16 index) + 16 B per extended unknown; measured to be bound by the per-level LDS dependency chain, not by bytes",
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return out


def cpu_baseline(D, pc, iters, n):
    """The oracle's C kernels (oracle/csrc/oracle_linalg.c: CSR SpMV, ILU(0) solve, CGS2) timed on ONE host core on
    the same matrices (exported from HBM): `iters` GMRES iterations.  kind = "port" (CPU restatement, not DAFoam)."""
    from oracle import linear as OL
    from dafoam_amd.pyDASolvers import Mat
    import ctypes as C
    from dafoam_amd import _capi

    t0 = time.time()
    # export the operator: re-assemble into a Mat handle to read it back
    A = Mat()
    D.solver.calcdRdWT(0, A, mode=1)
    Ah = A.to_scipy()
    Ph = pc.to_scipy()
    A.destroy()
    ilu = OL.ILU(Ph, fill=0)
    Ac = OL.CSR(Ah)
    rng = np.random.default_rng(1234)
    rhs = rng.standard_normal(n)
    prep = time.time() - t0
    t0 = time.perf_counter()
    x, info = OL.gmres(Ac.matvec, rhs, ilu.solve, restart=iters, fixed_iters=iters)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    Ac.matvec(rhs)
    t_spmv = time.perf_counter() - t1
    out = {
        "value": iters / dt,
        "unit": "iter/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{iters} GMRES iterations (oracle C SpMV + ILU(0) + CGS2, gcc -O3 -march=native) on the same dRdWT/dRdWTPC "
                  f"matrices copied back from HBM; one oracle SpMV = {t_spmv*1e3:.1f} ms; export+ILU prep {prep:.1f} s (untimed)",
        "spmv_GBps": (12.0 * Ah.nnz + 4.0 * (n + 1) + 16.0 * n) / t_spmv / 1e9,
    }
    # multi-core leg (informative, never allowed to break the line): chunked mat-vec + block-Jacobi ILU(0), one block per
    # thread - the reference's one-ASM-block-per-MPI-rank layout without the overlap
    try:
        if os.environ.get("DAS_BENCH_CPU_MT", "1") != "0":
            threads = max(2, min(32, (os.cpu_count() or 2)))
            t0 = time.time()
            T = OL.ThreadedOperators(Ah, Ph, threads, fill=0)
            prep_mt = time.time() - t0
            t0 = time.perf_counter()
            OL.gmres(T.matvec, rhs, T.pc_solve, restart=iters, fixed_iters=iters)
            dt_mt = time.perf_counter() - t0
            out["multicore"] = {"value": iters / dt_mt, "unit": "iter/s", "cores": T.threads, "kind": "port",
                                "sample": f"{iters} GMRES iterations, {T.threads} threads: row-chunked oracle SpMV + block-Jacobi ILU(0) "
                                          f"(one block per thread), serial CGS2; prep {prep_mt:.1f} s (untimed)"}
    except Exception as e:  # noqa: BLE001
        out["multicore"] = {"error": str(e)[:200]}
    return out


if __name__ == "__main__":
    main()
