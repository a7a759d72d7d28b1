"""GPU tier: two ranks of the sharded adjoint (dafoam_amd.distributed.ShardedAdjoint) on ONE MI355X (the test box
has a single GPU; transport = gloo with host staging - the production transport is nccl/RCCL) against the
single-domain GPU solve of the same global problem: the adjoint vector must agree to <= 1e-6."""
import os
import socket

import numpy as np
import pytest

from common import NORM_STATES, relerr

pytestmark = pytest.mark.gpu
NX, NY, NZ = 24, 8, 6
OPTS = {"solverName": "DASimpleFoam", "normalizeStates": dict(NORM_STATES),
        "adjEqnOption": {"gmresRelTol": 1e-11, "gmresMaxIters": 1500, "gmresRestart": 500, "printInfo": 0},
        "amd": {"pcBlockCells": 256}}


def _rhs_from_keys(key):
    return np.sin(0.37 * (key % 1009)) + 0.1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from dafoam_amd.distributed import ShardedAdjoint

        S = ShardedAdjoint(NX, NY, NZ, OPTS, device_index=0)
        S.setup()
        rhs = _rhs_from_keys(S.key)
        psi, fail = S.solve(rhs)
        info = S.ksp.info()
        q.put((rank, S.key[S.owned], psi[S.owned], fail, info["iters"], info["res"] / info["res0"], float(np.abs(psi[~S.owned]).max())))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_adjoint_matches_single_domain():
    import torch.multiprocessing as mp

    from dafoam_amd.distributed import SlabPartition, state_table
    from dafoam_amd.meshgen import bench_channel_case
    from dafoam_amd.pyDAFoam import PYDAFOAM

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single-domain solve of the same global problem
    gcase = bench_channel_case(NX, NY, NZ)
    gkey, _, _ = state_table(SlabPartition(NX, NY, NZ, 0, 1), gcase.mesh)
    D = PYDAFOAM(options=OPTS, case=gcase)
    psi_g, fail_g = D.solveAdjoint(_rhs_from_keys(gkey))
    assert fail_g == 0
    look = dict(zip(gkey.tolist(), range(gkey.size)))
    psi_s = np.full(gkey.size, np.nan)
    for rank, keys, psi, fail, iters, relres, ghostmax in res:
        assert fail == 0 and relres < 1e-9 and ghostmax == 0.0, (rank, fail, relres, ghostmax)
        psi_s[[look[k] for k in keys.tolist()]] = psi
    assert not np.isnan(psi_s).any()
    assert relerr(psi_s, psi_g) <= 1e-6
