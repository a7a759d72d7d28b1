// Multi-GPU communication of the adjoint solve (one process per GPU): halo REDUCTION of the ghost rows of dRdW^T.x and
// the all-reduce of the fused Gram-Schmidt dots, issued from C++ on HIP streams - no Python in the iteration loop.
//
// Reference communication sites: PETSc MatMult's VecScatter of the MPIAIJ off-diagonal block and
// the MPI_Allreduce of VecMDot/VecNorm inside KSPGMRES (DALinearEqn.C:341-437).  MI355X design:
//   * the ghost rows (extended states owned by a peer) are evaluated FIRST by a row-list SpMV that writes straight into
//     the send buffer, then grouped ncclSend/ncclRecv (RCCL over xGMI: every neighbour pair has its own link) run on a
//     communication stream WHILE the full-range SpMV of the owned rows runs on the compute stream; the received
//     contributions are added peer by peer (deterministic order) and the local ghost rows are zeroed;
//   * the (j+2) Hessenberg dots of an iteration travel as ONE in-stream ncclAllReduce.
// RCCL is bound at run time (dlopen of the copy PyTorch already loaded, else the system one): single-GPU runs never touch
// it.  The transport is replaceable by a host callback (das_set_exchange_cb) so that the pack / overlap / unpack logic is
// exercised by the gloo tests on boxes with one GPU.
#pragma once
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "das_common.hpp"

namespace das {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void load() {
        if (lib) return;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD); if (lib) break; }  // the copy already in the process
        for (const char* nm : names) { if (lib) break; lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); }
        DAS_CHECK(lib, DAS_ERR_INTERNAL, std::string("cannot load RCCL: ") + dlerror());
#define DAS_RCCL_SYM(field, name)                                                            \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name));                             \
    DAS_CHECK(field, DAS_ERR_INTERNAL, std::string("RCCL symbol missing: ") + name)
        DAS_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        DAS_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        DAS_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        DAS_RCCL_SYM(Send, "ncclSend");
        DAS_RCCL_SYM(Recv, "ncclRecv");
        DAS_RCCL_SYM(GroupStart, "ncclGroupStart");
        DAS_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        DAS_RCCL_SYM(AllReduce, "ncclAllReduce");
        DAS_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef DAS_RCCL_SYM
    }
};
inline RcclApi& rccl() {
    static RcclApi api;
    return api;
}
#define DAS_NCCL(expr)                                                                                                    \
    do {                                                                                                                  \
        ncclResult_t _r = (expr);                                                                                         \
        if (_r != ncclSuccess) throw das::Error(DAS_ERR_INTERNAL, std::string(#expr) + ": " + das::rccl().GetErrorString(_r)); \
    } while (0)

// sendBuf[k] = A[row rows[k], :] . x   (16 lanes per row; the ghost rows of the extended operator)
__global__ __launch_bounds__(256) void k_spmv_rows_to_buf(long long nrows, const int* __restrict__ rows, const long long* __restrict__ rp,
                                                          const int* __restrict__ ci, const double* __restrict__ v, const double* __restrict__ x,
                                                          double* __restrict__ buf) {
    const long long k = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    if (k >= nrows) return;
    const int row = rows[k];
    double acc = 0.0;
    for (long long q = rp[row] + lane; q < rp[row + 1]; q += 16) acc += v[q] * x[ci[q]];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_down(acc, o, 16);
    if (lane == 0) buf[k] = acc;
}
__global__ void k_halo_add(long long cnt, const int* __restrict__ idx, const double* __restrict__ buf, double* __restrict__ y) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) y[idx[k]] += buf[k];
}
__global__ void k_zero_idx(long long cnt, const int* __restrict__ idx, double* __restrict__ y) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) y[idx[k]] = 0.0;
}
__global__ void k_pack_idx(long long cnt, const int* __restrict__ idx, const double* __restrict__ y, double* __restrict__ buf) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) buf[k] = y[idx[k]];
}
__global__ void k_unpack_idx(long long cnt, const int* __restrict__ idx, const double* __restrict__ buf, double* __restrict__ y) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) y[idx[k]] = buf[k];
}

typedef void (*das_exchange_fn)(double* d_send, double* d_recv, void* user);

struct HaloPlan {
    bool active = false;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;         // native transport (null: callback transport / single rank)
    das_exchange_fn exchange_cb = nullptr;
    void* cb_user = nullptr;
    hipStream_t commStream = nullptr;
    hipEvent_t evPacked = nullptr, evDone = nullptr;
    std::vector<int> peers;
    std::vector<long long> sendOff, recvOff;  // npeers+1, offsets into the buffers (doubles)
    DevBuf<int> sendIdx, recvIdx, ghostIdx;
    DevBuf<double> sendBuf, recvBuf;
    long long nSend = 0, nRecv = 0, nGhost = 0;
    ~HaloPlan() {
        if (comm && rccl().CommDestroy) (void)rccl().CommDestroy(comm);
        if (evPacked) (void)hipEventDestroy(evPacked);
        if (evDone) (void)hipEventDestroy(evDone);
        if (commStream) (void)hipStreamDestroy(commStream);
    }
    void ensure_streams() {
        if (!commStream) DAS_HIP(hipStreamCreateWithFlags(&commStream, hipStreamNonBlocking));
        if (!evPacked) DAS_HIP(hipEventCreateWithFlags(&evPacked, hipEventDisableTiming));
        if (!evDone) DAS_HIP(hipEventCreateWithFlags(&evDone, hipEventDisableTiming));
    }
    // phase 1 (before the owned-row product): ghost rows -> send buffer, exchange started on the communication stream
    template <class MatT>
    void begin(const MatT& A, const double* x, hipStream_t st) {
        if (nSend > 0)
            hipLaunchKernelGGL(k_spmv_rows_to_buf, dim3((unsigned)((nSend + 15) / 16)), dim3(256), 0, st, nSend, sendIdx.p, A.rowptr.p, A.col.p, A.val.p, x,
                               sendBuf.p);
        if (comm) {
            DAS_HIP(hipEventRecord(evPacked, st));
            DAS_HIP(hipStreamWaitEvent(commStream, evPacked, 0));
            DAS_NCCL(rccl().GroupStart());
            for (size_t i = 0; i < peers.size(); i++) {
                const long long ns = sendOff[i + 1] - sendOff[i], nr = recvOff[i + 1] - recvOff[i];
                if (ns > 0) DAS_NCCL(rccl().Send(sendBuf.p + sendOff[i], (size_t)ns, ncclDouble, peers[i], comm, commStream));
                if (nr > 0) DAS_NCCL(rccl().Recv(recvBuf.p + recvOff[i], (size_t)nr, ncclDouble, peers[i], comm, commStream));
            }
            DAS_NCCL(rccl().GroupEnd());
            DAS_HIP(hipEventRecord(evDone, commStream));
        }
    }
    // phase 2 (after the owned-row product): received contributions added peer by peer, local ghost rows zeroed
    void finish(double* y, hipStream_t st) {
        if (comm) DAS_HIP(hipStreamWaitEvent(st, evDone, 0));
        else if (exchange_cb) exchange_cb(sendBuf.p, recvBuf.p, cb_user);  // host-staged transport (runs on `st` by contract)
        for (size_t i = 0; i < peers.size(); i++) {
            const long long nr = recvOff[i + 1] - recvOff[i];
            if (nr > 0) hipLaunchKernelGGL(k_halo_add, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, nr, recvIdx.p + recvOff[i], recvBuf.p + recvOff[i], y);
        }
        if (nGhost > 0) hipLaunchKernelGGL(k_zero_idx, dim3((unsigned)((nGhost + 255) / 256)), dim3(256), 0, st, nGhost, ghostIdx.p, y);
    }
    // the halo REDUCTION of a vector that is already evaluated on all extended rows (the sparse A Z product of the deflated coarse
    // mode): ghost-row values travel to their owners and are added there, the local ghost rows are zeroed
    void reduce_vector(double* y, hipStream_t st) {
        if (nSend > 0) hipLaunchKernelGGL(k_pack_idx, dim3((unsigned)((nSend + 255) / 256)), dim3(256), 0, st, nSend, sendIdx.p, (const double*)y, sendBuf.p);
        if (comm) {
            DAS_NCCL(rccl().GroupStart());
            for (size_t i = 0; i < peers.size(); i++) {
                const long long ns = sendOff[i + 1] - sendOff[i], nr = recvOff[i + 1] - recvOff[i];
                if (ns > 0) DAS_NCCL(rccl().Send(sendBuf.p + sendOff[i], (size_t)ns, ncclDouble, peers[i], comm, st));
                if (nr > 0) DAS_NCCL(rccl().Recv(recvBuf.p + recvOff[i], (size_t)nr, ncclDouble, peers[i], comm, st));
            }
            DAS_NCCL(rccl().GroupEnd());
        } else if (exchange_cb) exchange_cb(sendBuf.p, recvBuf.p, cb_user);
        for (size_t i = 0; i < peers.size(); i++) {
            const long long nr = recvOff[i + 1] - recvOff[i];
            if (nr > 0) hipLaunchKernelGGL(k_halo_add, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, nr, recvIdx.p + recvOff[i], recvBuf.p + recvOff[i], y);
        }
        if (nGhost > 0) hipLaunchKernelGGL(k_zero_idx, dim3((unsigned)((nGhost + 255) / 256)), dim3(256), 0, st, nGhost, ghostIdx.p, y);
    }
    // ---- overlap of the additive-Schwarz preconditioner (adjEqnOption.asmOverlap, reference DALinearEqn.C:212-216): the sub-domain
    // solve of a rank covers its owned unknowns plus `overlap` rings of ghost cells; before every apply the input vector's entries
    // on those ghost unknowns are GATHERED from their owners (the opposite direction of the reduction above, restricted to the overlap).
    // ovSendIdx: my owned states in a peer's overlap (packed in the peer's order), ovRecvIdx: my overlap ghost states, per peer.
    bool ovActive = false;
    std::vector<long long> ovSendOff, ovRecvOff;
    DevBuf<int> ovSendIdx, ovRecvIdx;
    DevBuf<double> ovSendBuf, ovRecvBuf;
    long long nOvSend = 0, nOvRecv = 0;
    das_exchange_fn gather_cb = nullptr;
    void* gather_user = nullptr;
    void gather_overlap(double* v, hipStream_t st) {
        if (!ovActive) return;
        if (nOvSend > 0) hipLaunchKernelGGL(k_pack_idx, dim3((unsigned)((nOvSend + 255) / 256)), dim3(256), 0, st, nOvSend, ovSendIdx.p, (const double*)v, ovSendBuf.p);
        if (comm) {
            DAS_NCCL(rccl().GroupStart());
            for (size_t i = 0; i < peers.size(); i++) {
                const long long ns = ovSendOff[i + 1] - ovSendOff[i], nr = ovRecvOff[i + 1] - ovRecvOff[i];
                if (ns > 0) DAS_NCCL(rccl().Send(ovSendBuf.p + ovSendOff[i], (size_t)ns, ncclDouble, peers[i], comm, st));
                if (nr > 0) DAS_NCCL(rccl().Recv(ovRecvBuf.p + ovRecvOff[i], (size_t)nr, ncclDouble, peers[i], comm, st));
            }
            DAS_NCCL(rccl().GroupEnd());
        } else if (gather_cb) gather_cb(ovSendBuf.p, ovRecvBuf.p, gather_user);
        else throw das::Error(DAS_ERR_STATE, "asmOverlap > 0 on several ranks needs a transport (RCCL communicator or das_set_gather_cb)");
        if (nOvRecv > 0) hipLaunchKernelGGL(k_unpack_idx, dim3((unsigned)((nOvRecv + 255) / 256)), dim3(256), 0, st, nOvRecv, ovRecvIdx.p, (const double*)ovRecvBuf.p, v);
    }
    void zero_overlap(double* z, hipStream_t st) {
        if (ovActive && nOvRecv > 0) hipLaunchKernelGGL(k_zero_idx, dim3((unsigned)((nOvRecv + 255) / 256)), dim3(256), 0, st, nOvRecv, ovRecvIdx.p, z);
    }
    // sum of a small device buffer over the ranks, in stream order
    bool allreduce(double* d_buf, int n, hipStream_t st) {
        if (!comm) return false;
        DAS_NCCL(rccl().AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, comm, st));
        return true;
    }
};

}  // namespace das
