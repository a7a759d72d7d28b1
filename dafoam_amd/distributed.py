"""Cell-partition sharding of the adjoint hot path across GPUs (one process per GPU, torch.distributed).

Reference: MPI domain decomposition (one OpenFOAM sub-domain per rank; processor patches + PETSc VecScatter,
SURVEY.md section 2.3).  MI355X design:

  * every rank holds an EXTENDED sub-mesh = its owned cells + GHOST_LAYERS (=3, the stencil depth of pRes,
    reference DAStateInfoSimpleFoam.C:86-93) layers of ghost cells; residuals of owned cells are exact on it;
  * each rank colours and assembles its own rows independently (colours are not shared between ranks);
    the assembled operator is A_ext^T : rows = all extended states, columns = owned residuals;
  * dRdW^T.x = local SpMV over the extended rows followed by ONE halo *reduction* (ghost-row contributions are
    sent to the owner rank and added there) - grouped point-to-point over xGMI (each neighbour pair has its own
    link); dots/norms = one small all-reduce per fused multi-dot;
  * the preconditioner: one node-block ILU per rank on its owned unknowns + adjEqnOption.asmOverlap rings of ghost cells (restricted
    additive Schwarz across the ranks like the reference's ASM, DALinearEqn.C:212-216; round 6): one owner -> ghost gather of the
    overlap entries per apply; ONE global pressure coarse space (one small all-reduce per apply).

The partition implemented here is a slab decomposition along x of the structured channel generators
(dafoam_amd.meshgen); the halo machinery itself only needs (owner rank, global key) per extended state.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from dataclasses import dataclass

import numpy as np

GHOST_LAYERS = 3
_KEY = 1 << 40


@dataclass
class SlabPartition:
    NX: int
    NY: int
    NZ: int
    rank: int
    world: int
    G: int = GHOST_LAYERS

    def __post_init__(self):
        assert self.NX % self.world == 0, "NX must be divisible by the number of ranks"
        w = self.NX // self.world
        assert w >= self.G, "slab thinner than the ghost depth"
        self.i0, self.i1 = self.rank * w, (self.rank + 1) * w
        self.e0, self.e1 = max(0, self.i0 - self.G), min(self.NX, self.i1 + self.G)
        self.nxl = self.e1 - self.e0

    def rank_of_column(self, gi):
        return gi // (self.NX // self.world)


def state_table(part: SlabPartition, mesh):
    """Per extended state (DAIndex 'state' ordering of DASimpleFoam): global key, owner rank (-1 = cut face: belongs to
    nobody), owned flag."""
    N, F, nIF = mesh.n_cells, mesh.n_faces, mesh.n_internal_faces
    nxl, NY, NX = part.nxl, part.NY, part.NX
    c = np.arange(N)
    il, j, k = c % nxl, (c // nxl) % NY, c // (nxl * NY)
    gi = il + part.e0
    cg = gi + NX * (j + NY * k)
    crank = part.rank_of_column(gi)
    own = mesh.owner.astype(np.int64)
    nei = mesh.neighbour.astype(np.int64)
    d = np.zeros(F, dtype=np.int64)
    diff = nei - own[:nIF]
    d[:nIF] = np.where(diff == 1, 0, np.where(diff == nxl, 1, 2))
    assert np.all((diff == 1) | (diff == nxl) | (diff == nxl * NY))
    is_cut = np.zeros(F, dtype=bool)
    for pi, p in enumerate(mesh.patches):
        sl = slice(p.start, p.start + p.size)
        d[sl] = 3 + pi
        if (pi == 0 and part.e0 > 0) or (pi == 1 and part.e1 < NX):
            is_cut[sl] = True
    fkey = (5 + d) * _KEY + cg[own]
    frank = np.where(is_cut, -1, crank[own])
    key = np.concatenate([0 * _KEY + np.repeat(cg, 3) * 3 + np.tile(np.arange(3), N), 3 * _KEY + cg, 4 * _KEY + cg, fkey])
    # U keys: kind 0 with 3*cg+comp keeps the three components distinct
    owner_rank = np.concatenate([np.repeat(crank, 3), crank, crank, frank])
    return key, owner_rank, owner_rank == part.rank


class HaloExchange:
    """Halo reduction of ghost-row contributions (torch tensors, CPU or GPU).  Works with any backend that has
    isend/irecv (nccl on GPUs = RCCL over xGMI; gloo in the CPU tests, staging GPU tensors through the host)."""

    def __init__(self, key, owner_rank, rank, world, device="cpu"):
        import torch
        import torch.distributed as dist

        self.rank, self.world = rank, world
        self.device = device
        self.n = key.size
        ghost = (owner_rank != rank)
        self.ghost_idx = torch.from_numpy(np.nonzero(ghost)[0]).to(device)
        # what I hold for others (ghost states owned by q), sorted by key
        send_keys = {}
        self.send_idx = {}
        for q in range(world):
            if q == rank:
                continue
            sel = np.nonzero(owner_rank == q)[0]
            if sel.size:
                o = np.argsort(key[sel], kind="stable")
                self.send_idx[q] = torch.from_numpy(sel[o]).to(device)
                send_keys[q] = key[sel][o]
        gathered = [None] * world
        dist.all_gather_object(gathered, send_keys)
        lookup = dict(zip(key.tolist(), range(key.size)))
        self.recv_idx = {}
        for q in range(world):
            if q == rank or rank not in gathered[q]:
                continue
            ks = gathered[q][rank]
            idx = np.fromiter((lookup[kk] for kk in ks.tolist()), dtype=np.int64, count=ks.size)
            assert np.all(owner_rank[idx] == rank), "peer lists a state as mine that I do not own"
            self.recv_idx[q] = torch.from_numpy(idx).to(device)
        self.peers = sorted(set(self.send_idx) | set(self.recv_idx))
        self.stage = dist.get_backend() == "gloo" and str(device) != "cpu"
        self.bytes_per_exchange = 8 * sum(int(v.numel()) for v in self.send_idx.values())

    def reduce_(self, w):
        """w (n,) : add ghost-row values into their owners, then zero the local ghost rows."""
        import torch
        import torch.distributed as dist

        ops, recv_bufs, keep = [], {}, []
        for q in self.peers:
            if q in self.send_idx:
                sb = w.index_select(0, self.send_idx[q])
                if self.stage:
                    sb = sb.cpu()
                keep.append(sb)
                ops.append(dist.P2POp(dist.isend, sb, q))
            if q in self.recv_idx:
                rb = torch.empty(self.recv_idx[q].numel(), dtype=w.dtype, device="cpu" if self.stage else w.device)
                recv_bufs[q] = rb
                ops.append(dist.P2POp(dist.irecv, rb, q))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        for q, rb in recv_bufs.items():
            w.index_add_(0, self.recv_idx[q], rb.to(w.device))
        if self.ghost_idx.numel():
            w.index_fill_(0, self.ghost_idx, 0.0)
        return w


def install_comm(obj, native=None):
    """Install the communication of one shard (obj: .h, .L, .key, .owner_rank, .rank, .world, .dev, .n, .halo):

      * backend nccl (RCCL, one GPU per rank): the halo plan goes to the C++ library, which issues grouped
        ncclSend/ncclRecv on its own communication stream overlapped with the owned-row product and an in-stream
        ncclAllReduce for the dots - nothing of torch.distributed runs inside the iteration loop (das_comm.hpp);
      * backend gloo (CPU tests / several ranks on one GPU): the SAME plan (pack, overlap order, unpack in C++) with a
        host-staged exchange callback, and the all-reduce callback.
    """
    import torch
    import torch.distributed as dist

    from . import _capi

    L, h, halo = obj.L, obj.h, obj.halo
    if native is None:
        native = dist.get_backend() == "nccl"
    peers = list(halo.peers)
    send = [halo.send_idx[q].cpu().numpy() if q in halo.send_idx else np.zeros(0, np.int64) for q in peers]
    recv = [halo.recv_idx[q].cpu().numpy() if q in halo.recv_idx else np.zeros(0, np.int64) for q in peers]
    sendOff = np.concatenate([[0], np.cumsum([a.size for a in send])]).astype(np.int64)
    recvOff = np.concatenate([[0], np.cumsum([a.size for a in recv])]).astype(np.int64)
    sendIdx = (np.concatenate(send) if peers else np.zeros(0)).astype(np.int32)
    recvIdx = (np.concatenate(recv) if peers else np.zeros(0)).astype(np.int32)
    ghostIdx = halo.ghost_idx.cpu().numpy().astype(np.int32)
    peers_a = np.asarray(peers, dtype=np.int32)
    ip, lp = _capi.c_int_p, _capi.c_ll_p
    if os.environ.get("DAFOAM_AMD_COMM", "") == "torch":
        native = False  # force the callback transport (torch.distributed issues the same RCCL calls)
    if native:
        # every rank must end up on the same transport, and ncclCommInitRank is collective: a rank that fails BEFORE it would
        # leave the others blocked inside.  Two agreed steps: (1) local - bind RCCL, rank 0 draws the unique id; all ranks
        # MIN-reduce the outcome through the torch backend; (2) only if every rank succeeded: the collective communicator
        # creation, agreed on the same way.  Any failure: all ranks fall back to the callback transport together.
        def agreed(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=obj.dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        ident = [None]
        ok = L.das_comm_load_rccl() == 0
        if ok and obj.rank == 0:
            buf = C.create_string_buffer(128)
            ok = L.das_comm_unique_id(buf) == 0
            ident[0] = buf.raw if ok else None
        ok = agreed(ok)
        if ok:
            dist.broadcast_object_list(ident, src=0)
            ok = agreed(L.das_comm_init_rccl(h, obj.rank, obj.world, ident[0]) == 0)
        if not ok:
            if obj.rank == 0:
                print("[dafoam_amd] native RCCL transport unavailable (%s); using the torch.distributed callback transport"
                      % L.das_last_error().decode(), file=sys.stderr, flush=True)
            _capi.check(L.das_comm_reset(h))
            native = False
    _capi.check(L.das_comm_set_halo(h, len(peers), peers_a.ctypes.data_as(ip), sendOff.ctypes.data_as(lp), sendIdx.ctypes.data_as(ip),
                                    recvOff.ctypes.data_as(lp), recvIdx.ctypes.data_as(ip), int(ghostIdx.size), ghostIdx.ctypes.data_as(ip)))
    obj._comm_native = bool(native)
    if native:
        obj._cb = ()
        return
    nS, nR = int(sendOff[-1]), int(recvOff[-1])
    stage = dist.get_backend() == "gloo"
    if stage:
        # gloo with device tensors = several ranks sharing ONE GPU (the 1-GPU test box): the per-XCD ticket counters of the preconditioner
        # sweeps assume workgroups of a launch resident on every XCD - not guaranteed when several processes compete for the compute units
        # (round 6: 8 ranks at 2 M cells stalled in 3 of 5 runs, 0 of 3 with the device-wide counter); read at the next factorisation
        os.environ.setdefault("DAS_BILU_XCD", "0")

    def exch_cb(ps, pr, _user):
        torch.cuda.current_stream(obj.dev).synchronize()
        sb = torch.as_tensor(_DevPtr(ps, nS), device=obj.dev) if nS else torch.empty(0, dtype=torch.float64, device=obj.dev)
        rb = torch.as_tensor(_DevPtr(pr, nR), device=obj.dev) if nR else None
        src = sb.cpu() if stage else sb
        dst = torch.empty(nR, dtype=torch.float64, device="cpu" if stage else obj.dev)
        ops = []
        for i, q in enumerate(peers):
            if sendOff[i + 1] > sendOff[i]:
                ops.append(dist.P2POp(dist.isend, src[sendOff[i]:sendOff[i + 1]], q))
            if recvOff[i + 1] > recvOff[i]:
                ops.append(dist.P2POp(dist.irecv, dst[recvOff[i]:recvOff[i + 1]], q))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        if nR:
            rb.copy_(dst)

    def ared_cb(ptr, m, _user):
        t = torch.as_tensor(_DevPtr(ptr, m), device=obj.dev)
        if stage:
            c = t.cpu()
            dist.all_reduce(c)
            t.copy_(c)
        else:
            dist.all_reduce(t)

    obj._cb = (_EXCH_CB(exch_cb), _ARED_CB(ared_cb))  # keep alive
    _capi.check(L.das_set_exchange_cb(h, C.cast(obj._cb[0], C.c_void_p), None))
    _capi.check(L.das_set_comm(h, None, C.cast(obj._cb[1], C.c_void_p), None))


def cell_adjacency(mesh):
    """Face-neighbour graph of the cells as a scipy CSR matrix: internal faces and coupled (cyclic) patch pairs."""
    import scipy.sparse as sp

    N, nIF = mesh.n_cells, mesh.n_internal_faces
    own, nei = np.asarray(mesh.owner, dtype=np.int64), np.asarray(mesh.neighbour, dtype=np.int64)
    rows, cols = [own[:nIF]], [nei]
    pname = {p.name: p for p in mesh.patches}
    for p in mesh.patches:
        if p.type == "cyclic" and p.size:
            q = pname[p.neighbour]
            rows.append(own[p.start : p.start + p.size])
            cols.append(own[q.start : q.start + q.size])
    r, c = np.concatenate(rows), np.concatenate(cols)
    A = sp.coo_matrix((np.ones(r.size, np.int8), (r, c)), shape=(N, N)).tocsr()
    return ((A + A.T) > 0).astype(np.int8).tocsr()


def overlap_mask(mesh, owner_rank, rank, overlap):
    """Unknowns of this rank's additive-Schwarz sub-domain (reference: PCASMSetOverlap(asmOverlap), DALinearEqn.C:212-216): the owned
    states plus the states anchored at the cells within `overlap` face-neighbour rings of the owned cells (a cell state is anchored at
    its cell, a face flux at the face's owner cell - the rule that assigns it to a rank); cut faces (owner_rank -1) belong to nobody.
    Returns a bool array over the extended states."""
    N, F = mesh.n_cells, mesh.n_faces
    n = owner_rank.size
    nsc = (n - 3 * N - F) // N
    assert n == (3 + nsc) * N + F and nsc >= 1
    inS = owner_rank[3 * N : 4 * N] == rank  # owner of the first scalar cell field = owner of the cell
    A = cell_adjacency(mesh)
    for _ in range(int(overlap)):
        inS = inS | (A @ inS.astype(np.int8) > 0)
    anchor = np.concatenate([np.repeat(np.arange(N), 3)] + [np.arange(N)] * nsc + [np.asarray(mesh.owner, dtype=np.int64)])
    return (inS[anchor] & (owner_rank >= 0)) | (owner_rank == rank)


def install_overlap(obj, overlap=None):
    """Restricted additive Schwarz across the ranks (das_set_pc_overlap): adjEqnOption.asmOverlap rings (default 1, like the
    reference; at most GHOST_LAYERS - 2 = 1 ring has the full PC stencil inside the extended sub-mesh, 2 rings are accepted) - the
    sub-domain mask, and per peer of the halo plan the lists of the owner -> ghost gather.  Collective.  Returns the overlap used."""
    import torch
    import torch.distributed as dist

    from . import _capi

    if overlap is None:
        adj = obj.D.getOption("adjEqnOption") if hasattr(obj.D, "getOption") else {}
        overlap = int(adj.get("asmOverlap", 1))
    amd = obj.D.getOption("amd") if hasattr(obj.D, "getOption") else {}
    overlap = max(0, min(int(overlap), GHOST_LAYERS - 1))
    L, h, halo = obj.L, obj.h, obj.halo
    obj.asm_overlap = 0
    if overlap == 0 or amd.get("pcType", "bilu") != "bilu":
        _capi.check(L.das_set_pc_overlap(h, None, 0, None, None, None, None))
        return 0
    mask = overlap_mask(obj.case.mesh, obj.owner_rank, obj.rank, overlap)
    peers = list(halo.peers)
    key, orank = obj.key, obj.owner_rank
    recv, want = [], {}
    for q in peers:  # my overlap ghost states owned by q, in key order: q packs in that order
        sel = np.nonzero(mask & (orank == q))[0]
        o = np.argsort(key[sel], kind="stable")
        recv.append(sel[o])
        want[q] = key[sel][o]
    gathered = [None] * obj.world
    dist.all_gather_object(gathered, want)
    lookup = dict(zip(key.tolist(), range(key.size)))
    send = []
    for q in peers:
        ks = gathered[q].get(obj.rank, np.zeros(0, np.int64))
        idx = np.fromiter((lookup[kk] for kk in ks.tolist()), dtype=np.int64, count=ks.size)
        assert np.all(orank[idx] == obj.rank), "peer asks for an overlap state I do not own"
        send.append(idx)
    sendOff = np.concatenate([[0], np.cumsum([a.size for a in send])]).astype(np.int64)
    recvOff = np.concatenate([[0], np.cumsum([a.size for a in recv])]).astype(np.int64)
    sendIdx = (np.concatenate(send) if peers else np.zeros(0)).astype(np.int32)
    recvIdx = (np.concatenate(recv) if peers else np.zeros(0)).astype(np.int32)
    m8 = np.ascontiguousarray(mask.astype(np.uint8))
    ip, lp = _capi.c_int_p, _capi.c_ll_p
    _capi.check(L.das_set_pc_overlap(h, m8.ctypes.data_as(C.POINTER(C.c_ubyte)), len(peers), sendOff.ctypes.data_as(lp), sendIdx.ctypes.data_as(ip),
                                     recvOff.ctypes.data_as(lp), recvIdx.ctypes.data_as(ip)))
    obj.asm_overlap, obj.pc_mask = overlap, mask
    obj.overlap_bytes_per_gather = 8 * int(sendOff[-1])
    if getattr(obj, "_comm_native", False):
        return overlap
    nS, nR = int(sendOff[-1]), int(recvOff[-1])
    stage = dist.get_backend() == "gloo"

    def gather_cb(ps, pr, _user):
        torch.cuda.current_stream(obj.dev).synchronize()
        sb = torch.as_tensor(_DevPtr(ps, nS), device=obj.dev) if nS else torch.empty(0, dtype=torch.float64, device=obj.dev)
        rb = torch.as_tensor(_DevPtr(pr, nR), device=obj.dev) if nR else None
        src = sb.cpu() if stage else sb
        dst = torch.empty(nR, dtype=torch.float64, device="cpu" if stage else obj.dev)
        ops = []
        for i, q in enumerate(peers):
            if sendOff[i + 1] > sendOff[i]:
                ops.append(dist.P2POp(dist.isend, src[sendOff[i]:sendOff[i + 1]], q))
            if recvOff[i + 1] > recvOff[i]:
                ops.append(dist.P2POp(dist.irecv, dst[recvOff[i]:recvOff[i + 1]], q))
        if ops:
            for r in dist.batch_isend_irecv(ops):
                r.wait()
        if nR:
            rb.copy_(dst)

    obj._gather_cb = _EXCH_CB(gather_cb)  # keep alive
    _capi.check(L.das_set_gather_cb(h, C.cast(obj._gather_cb, C.c_void_p), None))
    return overlap


def _forward_exchange(halo, w):
    """The reverse direction of HaloExchange.reduce_: every owner sends the values of its owned states to the ranks that hold
    them as ghosts (set-up time only: aggregate ids of the global coarse space)."""
    import torch
    import torch.distributed as dist

    ops, recv_bufs, keep = [], {}, []
    for q in halo.peers:
        if q in halo.recv_idx:  # my owned states that q holds as ghosts
            sb = w.index_select(0, halo.recv_idx[q])
            if halo.stage:
                sb = sb.cpu()
            keep.append(sb)
            ops.append(dist.P2POp(dist.isend, sb, q))
        if q in halo.send_idx:  # ghost states I hold that q owns
            rb = torch.empty(halo.send_idx[q].numel(), dtype=w.dtype, device="cpu" if halo.stage else w.device)
            recv_bufs[q] = rb
            ops.append(dist.P2POp(dist.irecv, rb, q))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    for q, rb in recv_bufs.items():
        w.index_copy_(0, halo.send_idx[q], rb.to(w.device))
    return w


class _DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


_HALO_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_ARED_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)
_EXCH_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p)


class ShardedAdjoint:
    """One rank of the sharded adjoint: extended-mesh case -> PYDAFOAM on this rank's GPU with owned mask, stream and
    communication callbacks installed."""

    def __init__(self, NX, NY, NZ, options, device_index=0, wall_function=False, state="prolonged", global_state=None,
                 case_kw=None):
        """global_state: optional (global_keys, global_W, global_yWall) - the extended states are then taken from a
        global state vector (e.g. a converged primal) instead of the prolonged fixture."""
        import torch
        import torch.distributed as dist

        from . import _capi
        from .meshgen import channel_case, load_coarse_primal, prolong_channel_state
        from .pyDAFoam import PYDAFOAM

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        part = self.part = SlabPartition(NX, NY, NZ, self.rank, self.world)
        co = load_coarse_primal()
        kw = dict(lengths=co["lengths"], grading_y=co["grading_y"])
        kw.update(case_kw or {})
        case = channel_case(part.nxl, NY, NZ, wall_function=wall_function, perturb=0.0, x_range=(part.e0, part.e1, NX), **kw)
        self.key, self.owner_rank, self.owned = state_table(part, case.mesh)
        if global_state is not None:
            gkey, gW, gy = global_state
            look = dict(zip(np.asarray(gkey).tolist(), range(len(gkey))))
            notcut = self.owner_rank >= 0
            gi = np.array([look.get(k, -1) for k in self.key.tolist()])
            assert np.all(gi[notcut] >= 0)
            W = case.states.copy()
            W[notcut] = np.asarray(gW)[gi[notcut]]
            nc = case.mesh.n_cells
            case.states = W
            case.y_wall = np.asarray(gy)[self.key[3 * nc : 4 * nc] - 3 * _KEY]
        elif state == "prolonged":
            prolong_channel_state(case, (part.nxl, NY, NZ), co, i0=part.e0, nx_global=NX)
        self.case = case
        opts = dict(options)
        opts["amdDevice"] = device_index
        self.D = PYDAFOAM(options=opts, case=case)
        L = self.L = _capi.lib()
        h = self.h = self.D.solver._h
        mask = np.ascontiguousarray(self.owned.astype(np.uint8))
        _capi.check(L.das_set_owned_mask(h, mask.ctypes.data_as(C.POINTER(C.c_ubyte))))
        self.dev = torch.device("cuda", device_index)
        self.halo = HaloExchange(self.key, self.owner_rank, self.rank, self.world, device=self.dev)
        _capi.check(L.das_set_stream(h, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        self.n = self.key.size
        _capi.check(L.das_set_n_global_cells(h, int(NX) * int(NY) * int(NZ)))
        install_comm(self)
        install_overlap(self)
        self.n_owned = int(self.owned.sum())

    # ------------------------------------------------------------------ solve_linear sequence on the shard
    def setup(self):
        from .pyDASolvers import KSP, Mat

        D = self.D
        D.solver.runColoring()
        self.pc = Mat()
        D.solver.calcdRdWT(1, self.pc)
        self.ksp = KSP()
        D.solverAD.createMLRKSPMatrixFree(self.pc, self.ksp)
        self.global_coarse = self.install_global_coarse()
        D.solverAD.initializedRdWTMatrixFree()

    def install_global_coarse(self):
        """ONE pressure coarse space over all ranks (das_ksp_set_global_coarse) instead of one per rank: the aggregates stay
        the per-rank RCB aggregates, numbered rank after rank; the ghost cells learn the aggregate of their owner through one
        owner -> ghost exchange of the p rows; E = Z^T P Z is summed over the ranks inside the library.  Returns the number of
        global aggregates (0: per-rank coarse spaces kept - amd.pcCoarseGlobal 0, no coarse space on some rank, or more than
        2048 aggregates in total)."""
        import torch
        import torch.distributed as dist

        amd = self.D.getOption("amd") if hasattr(self.D, "getOption") else {}
        if int(amd.get("pcCoarseGlobal", 1)) == 0 or amd.get("pcType", "bilu") != "bilu":
            return 0
        N = self.case.mesh.n_cells
        nloc, agg = self.ksp.coarse(N)
        stage = dist.get_backend() == "gloo"
        cnt = torch.zeros(self.world, dtype=torch.int64, device="cpu" if stage else self.dev)
        cnt[self.rank] = int(nloc)
        dist.all_reduce(cnt)
        cnt = cnt.cpu().numpy()
        if cnt.min() <= 0 or cnt.sum() > 2048:
            return 0
        off = int(cnt[: self.rank].sum())
        w = torch.full((self.n,), -1.0, dtype=torch.float64, device=self.dev)
        p0 = 3 * N  # the p block follows the velocity block in every solver with a pressure (DAIndex "state" ordering)
        gl = np.where(agg >= 0, agg + off, -1).astype(np.float64)
        w[p0 : p0 + N] = torch.from_numpy(gl).to(self.dev)
        _forward_exchange(self.halo, w)
        rows = np.ascontiguousarray(w[p0 : p0 + N].cpu().numpy().round().astype(np.int32))
        from . import _capi

        # the library validates the same conditions and would throw on ONE rank before the collective all-reduce of E, leaving the
        # peers blocked: agree on the validation result first
        ntot = int(cnt.sum())
        ok = off >= 0 and off + int(nloc) <= ntot and rows.min() >= -1 and rows.max() < ntot and bool(np.all(rows[agg >= 0] == agg[agg >= 0] + off))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cpu" if stage else self.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            return 0

        rc = _capi.check(self.L.das_ksp_set_global_coarse(self.h, self.ksp.handle, int(cnt.sum()), off, rows.ctypes.data_as(_capi.c_int_p)))
        return int(cnt.sum()) if rc == 0 else 0

    def solve(self, rhs_ext):
        """rhs_ext: extended-length vector (ghost entries ignored).  Returns (psi_ext with zero ghosts, fail)."""
        from .pyDASolvers import Vec

        b = Vec(self.n)
        b.array[:] = np.where(self.owned, rhs_ext, 0.0)
        x = Vec(self.n)
        fail = self.D.solverAD.solveLinearEqn(self.ksp, b, x)
        return x.array.copy(), fail

    def run_fixed(self, d_rhs, d_sol, iters):
        from . import _capi

        rc = self.L.das_ksp_run_fixed_device(self.h, self.ksp.handle, C.c_void_p(d_rhs.data_ptr()), C.c_void_p(d_sol.data_ptr()), int(iters))
        if rc < 0:
            raise _capi.DASError(self.L.das_last_error().decode())
        return rc


# =================================================================================================================
# General (unstructured) partitions: sub-mesh extraction with ghost rings
# =================================================================================================================
def rcb_partition(centres, nparts):
    """Recursive coordinate bisection of cell centres into `nparts` (power of two not required) balanced parts
    (the reference decomposes with scotch, pyDAFoam.py:597-604)."""
    part = np.zeros(len(centres), dtype=np.int32)

    def rec(idx, p0, np_):
        if np_ == 1:
            part[idx] = p0
            return
        nl = np_ // 2
        ext = centres[idx].max(0) - centres[idx].min(0)
        d = int(np.argmax(ext))
        o = np.argsort(centres[idx, d], kind="stable")
        k = len(idx) * nl // np_
        rec(idx[o[:k]], p0, nl)
        rec(idx[o[k:]], p0 + nl, np_ - nl)

    rec(np.arange(len(centres)), 0, nparts)
    return part


def preserve_patches(case, part, patch_names):
    """decomposeParDict.preservePatches (reference pyDAFoam.py:597-604, tests/runRegTests_DATurboFoamTransonic.py:67): both
    cells of every face pair of the named cyclic patches end up on ONE rank.  The cells connected through the pairs are merged
    into groups (union-find over the pair graph) and every group goes to the rank that owns most of its cells."""
    part = np.asarray(part).copy()
    m = case.mesh
    pname = {p.name: p for p in m.patches}
    parent = np.arange(m.n_cells)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    own = m.owner.astype(np.int64)
    for nm in patch_names:
        if nm in ("None", "none", ""):
            continue
        p = pname[nm]
        if p.type != "cyclic":
            continue
        q = pname[p.neighbour]
        for k in range(p.size):
            a, b = find(int(own[p.start + k])), find(int(own[q.start + k]))
            if a != b:
                parent[max(a, b)] = min(a, b)
    roots = np.array([find(c) for c in range(m.n_cells)])
    for r in np.unique(roots[roots != np.arange(m.n_cells)]):
        members = np.nonzero(roots == r)[0]
        part[members] = np.bincount(part[members]).argmax()
    return part


def extract_submesh(case, part, rank, G=GHOST_LAYERS):
    """Extended sub-mesh of `rank`: owned cells (part == rank) + G rings of ghost cells, as a FoamCase in OpenFOAM
    ordering.  Faces whose other cell lies outside the extended set become a trailing zero-gradient patch "ghostcut".
    Returns (sub_case, info) with info: cell_g (local->global cell), face_g (local->global face), face_sign (+1/-1, -1 if
    the local orientation is flipped), is_cut (local face), key / owner_rank / owned per extended state (global DAIndex
    index as key)."""
    import copy

    import scipy.sparse as sp

    from .meshgen import BC_ZERO_GRADIENT, NUT_CALCULATED, FoamCase, Patch, PolyMesh

    m = case.mesh
    N, F, nIF = m.n_cells, m.n_faces, m.n_internal_faces
    own, nei = m.owner.astype(np.int64), m.neighbour.astype(np.int64)
    # cell adjacency: internal faces AND coupled (cyclic) patch pairs - the paired cell is a face neighbour for every stencil
    # (csrc/das_mesh.cpp cyc_face), so the ghost rings run through the pair and both faces of a pair whose two cells are in the
    # extended set stay a cyclic pair of the sub-mesh.  A pair may be split between ranks (the partner is then a ghost cell);
    # the reference instead keeps pairs on one processor (decomposeParDict preservePatches, pyDAFoam.py:597-604: OpenFOAM's
    # processor-cyclic patches) - preserve_patches() below reproduces that for partitions that ask for it.
    pname = {p.name: p for p in m.patches}
    partner = np.full(F, -1, dtype=np.int64)  # cyclic face -> its paired face
    for p in m.patches:
        if p.type == "cyclic":
            q = pname[p.neighbour]
            assert q.size == p.size, "cyclic patches of a pair must have equal sizes"
            partner[p.start : p.start + p.size] = q.start + np.arange(p.size)
    cyc = np.nonzero(partner >= 0)[0]
    rows = np.concatenate([own[:nIF], own[cyc]])
    cols = np.concatenate([nei, own[partner[cyc]]])
    A = sp.coo_matrix((np.ones(rows.size, np.int8), (rows, cols)), shape=(N, N)).tocsr()
    A = ((A + A.T) > 0).astype(np.int8).tocsr()
    ext = part == rank
    for _ in range(G):
        ext = ext | (A @ ext.astype(np.int8) > 0)
    cell_g = np.nonzero(ext)[0]  # ascending global ids -> monotone renumbering keeps owner < neighbour
    loc = np.full(N, -1, dtype=np.int64)
    loc[cell_g] = np.arange(cell_g.size)
    o_in = ext[own]
    n_in = np.zeros(F, bool)
    n_in[:nIF] = ext[nei]
    f_int = np.nonzero(o_in[:nIF] & n_in[:nIF])[0]
    lo, ln = loc[own[f_int]], loc[nei[f_int]]
    order = np.lexsort((ln, lo))
    f_int = f_int[order]
    faces_g = [f_int]
    signs = [np.ones(f_int.size)]
    owners_l = [loc[own[f_int]]]
    patches = []
    start = f_int.size
    cut_b = []  # coupled faces whose partner cell is outside the extended set: cut like an internal face
    for p in m.patches:
        fs = np.arange(p.start, p.start + p.size)
        keep = o_in[fs]
        if p.type == "cyclic":
            both = keep & o_in[partner[fs]]
            cut_b.append(fs[keep & ~both])
            keep = both  # the same positions k survive in the partner patch: face k still pairs with face k
        fs = fs[keep]
        patches.append(Patch(p.name, p.type, start, fs.size, neighbour=p.neighbour, rotation=p.rotation))
        faces_g.append(fs)
        signs.append(np.ones(fs.size))
        owners_l.append(loc[own[fs]])
        start += fs.size
    cut_b = np.concatenate(cut_b) if cut_b else np.zeros(0, np.int64)
    # cut faces: internal global faces with exactly one cell inside the extended set
    cut_o = np.nonzero(o_in[:nIF] & ~n_in[:nIF])[0]  # inside cell is the global owner -> orientation kept
    cut_n = np.nonzero(~o_in[:nIF] & n_in[:nIF])[0]  # inside cell is the global neighbour -> flip
    fcut = np.concatenate([cut_o, cut_n, cut_b])
    scut = np.concatenate([np.ones(cut_o.size), -np.ones(cut_n.size), np.ones(cut_b.size)])
    ocut = np.concatenate([loc[own[cut_o]], loc[nei[cut_n]], loc[own[cut_b]]])
    oc = np.lexsort((fcut, ocut))
    patches.append(Patch("ghostcut", "patch", start, fcut.size))
    faces_g.append(fcut[oc])
    signs.append(scut[oc])
    owners_l.append(ocut[oc])
    face_g = np.concatenate(faces_g)
    face_sign = np.concatenate(signs)
    owner_l = np.concatenate(owners_l).astype(np.int32)
    is_cut = np.zeros(face_g.size, bool)
    is_cut[start:] = True
    # face vertex lists (flip orientation where needed); compact the points
    nv = np.diff(m.face_ptr)
    assert np.all(nv == nv[0]), "extract_submesh handles uniform polygons (hex meshes)"
    k = int(nv[0])
    fp = m.face_pts.reshape(F, k)[face_g].copy()
    flip = face_sign < 0
    fp[flip] = fp[flip][:, ::-1]
    used = np.unique(fp)
    pmap = np.full(m.n_points, -1, dtype=np.int64)
    pmap[used] = np.arange(used.size)
    sub = PolyMesh(points=m.points[used].copy(), face_ptr=(k * np.arange(face_g.size + 1)).astype(np.int32),
                   face_pts=np.ascontiguousarray(pmap[fp].ravel().astype(np.int32)), owner=owner_l,
                   neighbour=loc[nei[f_int]].astype(np.int32), patches=patches)
    # case data
    sc = copy.copy(case)
    sc.mesh = sub
    sc.bcs = dict(case.bcs)
    sc.bcs["ghostcut"] = {"U": (BC_ZERO_GRADIENT, (0.0, 0.0, 0.0)), "p": (BC_ZERO_GRADIENT, 0.0), "T": (BC_ZERO_GRADIENT, 0.0),
                          "nuTilda": (BC_ZERO_GRADIENT, 0.0), "nut": (NUT_CALCULATED, 0.0)}
    sc.y_wall = None if case.y_wall is None else case.y_wall[cell_g]
    if getattr(case, "mrf", None):  # cut faces are interior faces of the zone: relative flux as for internal faces
        sc.mrf = dict(case.mrf)
        sc.mrf["nonRotatingPatches"] = list(case.mrf.get("nonRotatingPatches", ())) + ["ghostcut"]
    nl = cell_g.size
    W = case.states
    solver = case.solver_name
    nsc = {"DASimpleFoam": 3 if getattr(case, "has_T", False) else 2, "DARhoSimpleFoam": 3, "DATurboFoam": 3}[solver]
    blocks_g = [np.repeat(3 * cell_g, 3) + np.tile(np.arange(3), nl)] + [(3 + b) * N + cell_g for b in range(nsc)] + [(3 + nsc) * N + face_g]
    key = np.concatenate(blocks_g)
    sgn = np.concatenate([np.ones(key.size - face_g.size), face_sign])
    sc.states = W[key] * sgn
    crank = part[cell_g]
    frank = np.where(is_cut, -1, part[own[face_g]])
    owner_rank = np.concatenate([np.repeat(crank, 3)] + [crank] * nsc + [frank])
    info = dict(cell_g=cell_g, face_g=face_g, face_sign=face_sign, is_cut=is_cut, key=key, owner_rank=owner_rank, owned=owner_rank == rank,
                state_sign=sgn)
    return sc, info


class ShardedAdjointGeneral(ShardedAdjoint):
    """Sharded adjoint for an ARBITRARY global case and cell partition vector.  Two ways in:
      * ShardedAdjointGeneral(global_case, part, options): every rank holds the global case (e.g. read with
        dafoam_amd.foam_io.read_case) and extracts its own extended sub-mesh;
      * ShardedAdjointGeneral.scattered(global_case | None, part | None, options): only the SOURCE rank holds the global case; it
        extracts the extended sub-mesh of every rank and scatters them (the reference's decomposePar step, pyDAFoam.py:597-604,
        done in memory) - the other ranks never see more than their own cells plus three ghost rings.
    Then it proceeds like ShardedAdjoint."""

    def __init__(self, global_case, part, options, device_index=0):
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        assert int(np.asarray(part).max()) + 1 <= world
        case, info = extract_submesh(global_case, np.asarray(part), rank)
        self._init_from_sub(case, info, int(global_case.mesh.n_cells), options, device_index)

    @classmethod
    def scattered(cls, global_case, part, options, device_index=0, src=0):
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        subs = None
        if rank == src:
            assert int(np.asarray(part).max()) + 1 <= world
            subs = [extract_submesh(global_case, np.asarray(part), r) + (int(global_case.mesh.n_cells),) for r in range(world)]
        out = [None]
        dist.scatter_object_list(out, subs, src=src)
        case, info, nglobal = out[0]
        obj = cls.__new__(cls)
        obj._init_from_sub(case, info, nglobal, options, device_index)
        return obj

    def _init_from_sub(self, case, info, n_global_cells, options, device_index):
        import torch
        import torch.distributed as dist

        from . import _capi
        from .pyDAFoam import PYDAFOAM

        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.case, self.info = case, info
        self.key, self.owner_rank, self.owned = info["key"], info["owner_rank"], info["owned"]
        opts = dict(options)
        opts["amdDevice"] = device_index
        self.D = PYDAFOAM(options=opts, case=case)
        L = self.L = _capi.lib()
        h = self.h = self.D.solver._h
        mask = np.ascontiguousarray(self.owned.astype(np.uint8))
        _capi.check(L.das_set_owned_mask(h, mask.ctypes.data_as(C.POINTER(C.c_ubyte))))
        self.dev = torch.device("cuda", device_index)
        self.halo = HaloExchange(self.key, self.owner_rank, self.rank, self.world, device=self.dev)
        _capi.check(L.das_set_stream(h, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)))
        self.n = self.key.size
        _capi.check(L.das_set_n_global_cells(h, int(n_global_cells)))
        install_comm(self)
        install_overlap(self)
        self.n_owned = int(self.owned.sum())
