"""CPU tier, world_size 2 over gloo: the sharding logic of dafoam_amd.distributed (slab partition with 3 ghost
layers, ownership, global keys, halo reduction) checked with the ORACLE as the local operator:
  * residuals of owned cells evaluated on the extended sub-mesh == global residual (ghost depth = stencil depth),
  * owned states of the two ranks partition the global state vector,
  * local (A_ext^T restricted to owned columns) @ psi followed by the halo reduction == global dRdW^T psi."""
import os
import socket

import numpy as np
import pytest

from common import NORM_STATES

NX, NY, NZ = 8, 4, 3


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
        from dafoam_amd.distributed import HaloExchange, SlabPartition, state_table
        from dafoam_amd.meshgen import channel_case
        from oracle import jacobian as J
        from oracle.foam_mesh import Geometry
        from oracle.residual import residual

        kw = dict(lengths=(2.0, 0.2, 0.2), grading_y=2.0, perturb=0.0)
        gcase = channel_case(NX, NY, NZ, **kw)
        gpart = SlabPartition(NX, NY, NZ, 0, 1)
        gkey, _, _ = state_table(gpart, gcase.mesh)
        glook = dict(zip(gkey.tolist(), range(gkey.size)))
        part = SlabPartition(NX, NY, NZ, rank, world)
        case = channel_case(part.nxl, NY, NZ, x_range=(part.e0, part.e1, NX), **kw)
        key, orank, owned = state_table(part, case.mesh)
        # states / wall distance of the extended mesh taken from the global case (exact consistency)
        notcut = orank >= 0
        gidx = np.array([glook.get(k, -1) for k in key.tolist()])
        assert np.all(gidx[notcut] >= 0)
        W = case.states.copy()
        W[notcut] = gcase.states[gidx[notcut]]
        ncell = case.mesh.n_cells
        ckey = key[3 * ncell : 4 * ncell] - 3 * (1 << 40)
        case.y_wall = gcase.y_wall[ckey]
        case.states = W
        g = Geometry(case.mesh)
        gg = Geometry(gcase.mesh)
        # 1. owned residual rows are exact
        R = residual(case, g, W)
        Rg = residual(gcase, gg, gcase.states)
        err_res = np.abs(R[owned] - Rg[gidx[owned]]).max() / np.abs(Rg).max()
        # 2. sharded product
        sc = J.state_scales(case, g, NORM_STATES)
        con = J.connectivity(case, g)
        col, _ = J.greedy_coloring(con)
        A = J.jacobian_colored(case, g, W, con, col, sc, mode="cs", lower_bound=0).tocsc()
        A = A[:, np.nonzero(owned)[0]]  # columns = owned residuals
        scg = J.state_scales(gcase, gg, NORM_STATES)
        cong = J.connectivity(gcase, gg)
        colg, _ = J.greedy_coloring(cong)
        Ag = J.jacobian_colored(gcase, gg, gcase.states, cong, colg, scg, mode="cs", lower_bound=0)
        psi_g = np.random.default_rng(0).standard_normal(gkey.size)
        psi_owned = psi_g[gidx[owned]]
        w = torch.from_numpy(A @ psi_owned)
        halo = HaloExchange(key, orank, rank, world)
        halo.reduce_(w)
        w = w.numpy()
        ref = (Ag @ psi_g)[gidx[owned]]
        err_prod = np.abs(w[owned] - ref).max() / np.abs(ref).max()
        ghost_zero = float(np.abs(w[~owned]).max()) if (~owned).any() else 0.0
        nown = torch.tensor([int(owned.sum())])
        dist.all_reduce(nown)
        q.put((rank, err_res, err_prod, ghost_zero, int(nown.item()), gkey.size, halo.bytes_per_exchange))
    except Exception as e:  # a crashed worker must fail the test at once, not after the queue timeout
        import traceback

        q.put(("error", rank, traceback.format_exc()[-1500:]))
        raise
    finally:
        dist.destroy_process_group()


def test_sharded_product_matches_global_oracle():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        item = q.get(timeout=600)
        assert item[0] != "error", item
        res.append(item)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err_res, err_prod, ghost_zero, nown, nglob, nbytes in res:
        assert err_res < 1e-12, (rank, err_res)
        assert err_prod < 1e-11, (rank, err_prod)
        assert ghost_zero == 0.0
        assert nown == nglob
        assert nbytes > 0


def _worker_general(rank, world, port, q, kind="simple"):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dafoam_amd.distributed import HaloExchange, extract_submesh, rcb_partition
        from dafoam_amd.meshgen import channel_case, renumber_case
        from oracle import jacobian as J
        from oracle.foam_mesh import Geometry
        from oracle.residual import residual

        from common import norm_states
        from dafoam_amd.meshgen import rho_channel_case

        if kind == "rho":  # BASELINE configs[3]: DARhoSimpleFoam, 4-way cell partition
            gcase = rho_channel_case(NX, NY, NZ + 1, lengths=(2.0, 0.2, 0.2), grading_y=2.0)  # (renumber_case knows the incompressible layout only)
        elif kind == "wing":  # the default bench workload at N > 1: NACA0012 wing, spanwise slabs of whole cell layers (bench.py)
            from dafoam_amd.meshgen import naca0012_case

            gcase = naca0012_case(16, 5, 8, span=0.8)
        elif kind == "wing_columns":
            from dafoam_amd.meshgen import naca0012_case

            gcase = naca0012_case(16, 6, 4, span=0.4, fold_seam=True)
        else:
            gcase = renumber_case(channel_case(NX, NY, NZ, lengths=(2.0, 0.2, 0.2), grading_y=2.0, perturb=0.0), seed=11)
        NS = norm_states(gcase)
        gg = Geometry(gcase.mesh)
        if kind == "wing":
            layer = np.arange(gcase.mesh.n_cells, dtype=np.int64) // (16 * 5)
            assert np.allclose(gg.C[layer == 3][:, 2], gg.C[layer == 3][0, 2])  # the generator numbers the cells layer by layer
            part = (layer * world // 8).astype(np.int32)
        elif kind == "wing_columns":
            # round 6: what `bench.py --gpus N` does by default - the O-grid numbered across its seam (fold_seam), cut into blocks of the
            # (around, wall-normal) index plane that keep whole spanwise columns of cells (bench.naca_partition decodes the ring position)
            import sys

            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from bench import naca_partition
            from dafoam_amd.meshgen import naca_ring_position

            cid = np.arange(gcase.mesh.n_cells, dtype=np.int64)
            part = naca_partition(cid, (16, 6, 4), world, "columns", fold=True)
            assert np.unique(part).size == world
            i, j = naca_ring_position(cid, 16, True), (cid // 16) % 6
            for r in range(world):  # every rank: a rectangle of the index plane, all 4 layers of each of its columns
                sel = part == r
                cols = np.unique(i[sel] * 100 + j[sel])
                assert sel.sum() == cols.size * 4
                assert (i[sel].max() - i[sel].min() + 1) * (j[sel].max() - j[sel].min() + 1) == cols.size
        else:
            part = rcb_partition(gg.C, world)
        # the sub-meshes are extracted on rank 0 and scattered (what ShardedAdjointGeneral.scattered does)
        subs = [extract_submesh(gcase, part, r) for r in range(world)] if rank == 0 else None
        out = [None]
        dist.scatter_object_list(out, subs, src=0)
        case, info = out[0]
        g = Geometry(case.mesh)
        owned, key = info["owned"], info["key"]
        R = residual(case, g, case.states)
        Rg = residual(gcase, gg, gcase.states)
        err_res = np.abs(R[owned] - Rg[key[owned]]).max() / np.abs(Rg).max()
        sc = J.state_scales(case, g, NS)
        con = J.connectivity(case, g)
        col, _ = J.greedy_coloring(con)
        A = J.jacobian_colored(case, g, case.states, con, col, sc, mode="cs", lower_bound=0).tocsc()[:, np.nonzero(owned)[0]]
        scg = J.state_scales(gcase, gg, NS)
        cong = J.connectivity(gcase, gg)
        colg, _ = J.greedy_coloring(cong)
        Ag = J.jacobian_colored(gcase, gg, gcase.states, cong, colg, scg, mode="cs", lower_bound=0)
        psi_g = np.random.default_rng(0).standard_normal(gcase.states.size)
        w = torch.from_numpy(A @ psi_g[key[owned]])
        halo = HaloExchange(key, info["owner_rank"], rank, world)
        halo.reduce_(w)
        w = w.numpy()
        ref = (Ag @ psi_g)[key[owned]]
        q.put((rank, err_res, np.abs(w[owned] - ref).max() / np.abs(ref).max(), int(owned.sum()), gcase.states.size))
    except Exception as e:  # a crashed worker must fail the test at once, not after the queue timeout
        import traceback

        q.put(("error", rank, traceback.format_exc()[-1500:]))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "simple"), (4, "simple"), (4, "rho"), (2, "wing"), (4, "wing_columns")])
def test_general_partition_unstructured_mesh(world, kind):
    """Arbitrary (RCB) partition of a randomly renumbered mesh on 2 and 4 ranks, DASimpleFoam and DARhoSimpleFoam, and the NACA0012 wing cut
    into spanwise slabs of whole layers (what `bench.py --gpus N` does with its default workload, round 5): extract_submesh (on rank 0,
    scattered) + halo reduction == global oracle."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_general, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        item = q.get(timeout=600)
        assert item[0] != "error", item
        res.append(item)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sum(r[3] for r in res) == res[0][4]
    for rank, err_res, err_prod, nown, nglob in res:
        assert err_res < 1e-11 and err_prod < (1e-9 if kind == "rho" else 1e-11), (rank, err_res, err_prod)
