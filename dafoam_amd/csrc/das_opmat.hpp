// Packed operator format for dRdW^T.psi (reference DASolver::dRdWTMatVecMultFunction, DASolver.C:1364-1409).
//
// In the transposed CSR the three rows of a VECTOR state of one cell (U_x, U_y, U_z of cell c: rows 3c, 3c+1, 3c+2) have the
// SAME column list - the connectivity treats a vector state as three components that are connected together
// (DAJacCon::addStateConnections, DAJacCon.C:436-446) - and those rows hold 70 % of all entries (247 of the 1059 entries per
// cell are U-rows, three times).  The scalar CSR streams that list three times and gathers the same psi entries three times.
// Here the vector rows are stored as GROUP rows: one shared column list, three value planes, in chunks of 16 entries
//     chunk = [ int32 col[16] | double val[3][16] ]   (448 bytes, 64-byte aligned; the last chunk of a row is padded with
//                                                      zero values and a repeated valid column)
// so that the 16 lanes of a row read one contiguous 64-byte index segment and three contiguous 128-byte value segments per
// step, gather psi once and feed three accumulators: 28 bytes per 3 entries instead of 36, one gather instead of three.
// The scalar rows (p, nuTilda, phi, ...) keep the CSR arrays of the assembled matrix (no copy).  Algorithmic bytes at 2 M cells:
// 25.4 GB (12 B/entry CSR) -> 22.0 GB.  Built once per operator from the assembled CSR (one pass, ~10 ms); falls back to the
// scalar kernel when the three rows of a cell do not share their list (jacLowerBounds compaction).
#pragma once
#include "das_common.hpp"

namespace das {

constexpr int VP_CHUNK = 16;
constexpr int VP_CHUNK_BYTES = 64 + 3 * 16 * 8;  // 448

struct VecPack {
    bool ready = false;
    long long nGroups = 0, nChunks = 0, row0 = 0;  // vector rows are row0 + 3 g + d
    long long csrEntries = 0;                      // entries of the CSR rows this pack replaces
    DevBuf<long long> cptr;                        // nGroups + 1 chunk offsets
    DevBuf<unsigned char> data;                    // nChunks * VP_CHUNK_BYTES
    long long bytes() const { return nChunks * (long long)VP_CHUNK_BYTES + (nGroups + 1) * 8; }
    void release() { cptr.release(); data.release(); ready = false; nGroups = nChunks = csrEntries = 0; }
};

// chunks per group row, and: do the three rows share one column list?  (16 lanes per group)
__global__ __launch_bounds__(256) void k_vecpack_count(long long nG, long long row0, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                       int* __restrict__ nch, int* __restrict__ mismatch) {
    const long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    if (g >= nG) return;
    const long long r = row0 + 3 * g;
    const long long b0 = rp[r], b1 = rp[r + 1], b2 = rp[r + 2], e2 = rp[r + 3];
    const long long len = b1 - b0;
    bool bad = (b2 - b1 != len) || (e2 - b2 != len);
    if (!bad)
        for (long long k = lane; k < len; k += 16) {
            const int c = ci[b0 + k];
            bad |= (ci[b1 + k] != c) || (ci[b2 + k] != c);
        }
    if (bad) atomicOr(mismatch, 1);
    if (lane == 0) nch[g] = (int)((len + VP_CHUNK - 1) / VP_CHUNK);
}
__global__ __launch_bounds__(256) void k_vecpack_fill(long long nG, long long row0, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                      const double* __restrict__ v, const long long* __restrict__ cptr, unsigned char* __restrict__ data) {
    const long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    if (g >= nG) return;
    const long long r = row0 + 3 * g;
    const long long b0 = rp[r], b1 = rp[r + 1], b2 = rp[r + 2];
    const long long len = b1 - b0;
    const long long c0 = cptr[g], c1 = cptr[g + 1];
    for (long long ch = c0; ch < c1; ch++) {
        const long long k = (ch - c0) * VP_CHUNK + lane;
        const bool in = k < len;
        unsigned char* base = data + ch * VP_CHUNK_BYTES;
        reinterpret_cast<int*>(base)[lane] = ci[b0 + (in ? k : len - 1)];
        double* vv = reinterpret_cast<double*>(base + 64);
        vv[lane] = in ? v[b0 + k] : 0.0;
        vv[16 + lane] = in ? v[b1 + k] : 0.0;
        vv[32 + lane] = in ? v[b2 + k] : 0.0;
    }
}

// y[row0 + 3g + d] = sum_k val[d][k] x[col[k]]: 16 lanes per group row, VP_UNROLL chunks in flight per row
#ifndef VP_UNROLL
#define VP_UNROLL 4
#endif
__global__ __launch_bounds__(256) void k_spmv_vec3(long long nG, long long row0, const long long* __restrict__ cptr, const unsigned char* __restrict__ data,
                                                   const double* __restrict__ x, double* __restrict__ y) {
    const long long g = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    if (g >= nG) return;
    const long long c0 = cptr[g], c1 = cptr[g + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    long long ch = c0;
    for (; ch + VP_UNROLL <= c1; ch += VP_UNROLL) {
        int cc[VP_UNROLL];
        double v0[VP_UNROLL], v1[VP_UNROLL], v2[VP_UNROLL];
#pragma unroll
        for (int u = 0; u < VP_UNROLL; u++) {
            const unsigned char* base = data + (ch + u) * VP_CHUNK_BYTES;
            cc[u] = reinterpret_cast<const int*>(base)[lane];
            const double* vv = reinterpret_cast<const double*>(base + 64);
            v0[u] = vv[lane]; v1[u] = vv[16 + lane]; v2[u] = vv[32 + lane];
        }
#pragma unroll
        for (int u = 0; u < VP_UNROLL; u++) {
            const double xx = x[cc[u]];
            a0 += v0[u] * xx; a1 += v1[u] * xx; a2 += v2[u] * xx;
        }
    }
    if (ch < c1) {  // tail: the remaining 1..VP_UNROLL-1 chunks requested together (clamped to the last chunk, masked)
        int cc[VP_UNROLL];
        double v0[VP_UNROLL], v1[VP_UNROLL], v2[VP_UNROLL];
#pragma unroll
        for (int u = 0; u < VP_UNROLL - 1; u++) {
            const bool in = ch + u < c1;
            const unsigned char* base = data + (in ? ch + u : c1 - 1) * VP_CHUNK_BYTES;
            cc[u] = reinterpret_cast<const int*>(base)[lane];
            const double* vv = reinterpret_cast<const double*>(base + 64);
            v0[u] = in ? vv[lane] : 0.0; v1[u] = in ? vv[16 + lane] : 0.0; v2[u] = in ? vv[32 + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < VP_UNROLL - 1; u++) {
            const double xx = x[cc[u]];
            a0 += v0[u] * xx; a1 += v1[u] * xx; a2 += v2[u] * xx;
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 16);
        a1 += __shfl_down(a1, o, 16);
        a2 += __shfl_down(a2, o, 16);
    }
    if (lane == 0) {
        double* yo = y + row0 + 3 * g;
        yo[0] = a0; yo[1] = a1; yo[2] = a2;
    }
}

// build the packed vector rows [row0, row0 + 3 nG) of the CSR (rp, ci, v); returns false (and leaves P empty) when the rows of
// a cell do not share their column list
inline bool vecpack_build(VecPack& P, long long nG, long long row0, const long long* d_rp, const int* d_ci, const double* d_v, hipStream_t st) {
    P.release();
    if (nG <= 0) return false;
    DevBuf<int> nch(nG), bad(1);
    DAS_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_vecpack_count, dim3((unsigned)((nG + 15) / 16)), dim3(256), 0, st, nG, row0, d_rp, d_ci, nch.p, bad.p);
    DAS_HIP(hipStreamSynchronize(st));
    if (bad.to_host()[0]) return false;
    std::vector<int> h = nch.to_host();
    std::vector<long long> cp(nG + 1, 0);
    for (long long g = 0; g < nG; g++) cp[g + 1] = cp[g] + h[g];
    P.nGroups = nG; P.row0 = row0; P.nChunks = cp[nG];
    {
        long long e[2] = {0, 0};
        DAS_HIP(hipMemcpy(&e[0], d_rp + row0, sizeof(long long), hipMemcpyDeviceToHost));
        DAS_HIP(hipMemcpy(&e[1], d_rp + row0 + 3 * nG, sizeof(long long), hipMemcpyDeviceToHost));
        P.csrEntries = e[1] - e[0];
    }
    P.cptr.upload(cp);
    P.data.alloc((size_t)std::max<long long>(1, P.nChunks) * VP_CHUNK_BYTES);
    hipLaunchKernelGGL(k_vecpack_fill, dim3((unsigned)((nG + 15) / 16)), dim3(256), 0, st, nG, row0, d_rp, d_ci, d_v, P.cptr.p, P.data.p);
    DAS_HIP(hipStreamSynchronize(st));
    P.ready = true;
    return true;
}

}  // namespace das
