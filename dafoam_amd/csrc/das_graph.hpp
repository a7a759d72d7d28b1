// Graph set-up on the device (round 3): the transposed connectivity structures of dRdWT / dRdWTPC and the column -> net
// incidence of the colouring, derived from the row-major pattern the host builds (das_jaccon.cpp) instead of by host passes over
// 2.1 G + 1.4 G entries (chunked counting transposes: 3.7 s, CSC of the kept rows: 1.3 s, their upload: ~1 s at 2 M cells; all of
// it host-memory bound).  Reference: DAJacCon::setupdRdWCon builds the pattern (DAJacCon.C:2039-2600), DAPartDeriv inserts
// transposed (DAPartDeriv.C:192-201); this file is the "transpose" half, on the GPU:
//   1. count the entries of every column (atomicAdd on n counters), exclusive scan -> t_rowptr
//   2. fill with per-column cursors (arrival order), then SORT every transposed row ascending (bitonic in LDS): the structure is
//      the same as the host's, bit for bit (tests compare), and the three rows of a vector state share their list again
//   3. colouring input: the transposed structure restricted to the kept rows ("nets"), with the position of the column inside
//      each net found by binary search in the (ascending) row-major row
#pragma once
#include <vector>

#include "das_common.hpp"

namespace das {

// ---- exclusive scan of n int counts into n+1 long long offsets (two passes over blocks of 1024) -------------------------
__global__ __launch_bounds__(256) void k_scan_block_sums(long long n, const int* __restrict__ cnt, long long* __restrict__ bsum) {
    __shared__ long long red[4];
    const long long base = (long long)blockIdx.x * 1024;
    long long a = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) { const long long i = base + threadIdx.x + 256 * t; if (i < n) a += cnt[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// every block rewrites its 1024 counts as offsets: thread t owns 4 consecutive entries
__global__ __launch_bounds__(256) void k_scan_apply(long long n, const int* __restrict__ cnt, const long long* __restrict__ boff, long long* __restrict__ out) {
    __shared__ long long wsum[4];
    const long long base = (long long)blockIdx.x * 1024 + 4LL * threadIdx.x;
    long long c[4], a = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) { c[t] = base + t < n ? cnt[base + t] : 0; a += c[t]; }
    // inclusive scan of `a` over the 256 threads: within a wave by shuffles, then across the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long incl = a;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    long long run = boff[blockIdx.x] + woff + incl - a;
#pragma unroll
    for (int t = 0; t < 4; t++) { if (base + t < n) out[base + t] = run; run += c[t]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = boff[blockIdx.x] + wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// out[0..n] = exclusive prefix of cnt[0..n); returns the total
inline long long device_exclusive_scan(long long n, const int* d_cnt, long long* d_out, hipStream_t st) {
    const long long nb = (n + 1023) / 1024;
    DevBuf<long long> bsum((size_t)nb), boff((size_t)nb);
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(256), 0, st, n, d_cnt, bsum.p);
    std::vector<long long> h((size_t)nb);
    DAS_HIP(hipMemcpyAsync(h.data(), bsum.p, nb * sizeof(long long), hipMemcpyDeviceToHost, st));
    DAS_HIP(hipStreamSynchronize(st));
    long long run = 0;
    for (long long b = 0; b < nb; b++) { const long long v = h[b]; h[b] = run; run += v; }
    DAS_HIP(hipMemcpyAsync(boff.p, h.data(), nb * sizeof(long long), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, st, n, d_cnt, boff.p, d_out);
    DAS_HIP(hipStreamSynchronize(st));
    return run;
}

// ---- transpose ---------------------------------------------------------------------------------------------------------
// (grid-stride: a launch may not exceed 2^32 threads - one thread per entry failed silently beyond 4.29e9 entries, i.e. above ~4 M cells:
//  the first thing the 5 M-cell capacity run of round 5 hit, profiles/r06m_capacity_*)
__global__ __launch_bounds__(256) void k_tr_count(long long nnz, const int* __restrict__ col, int* __restrict__ cnt) {
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) atomicAdd(&cnt[col[k]], 1);
}
// 16 lanes per row: entry (r, j) goes to the next free slot of transposed row j
__global__ __launch_bounds__(256) void k_tr_fill(long long n, const long long* __restrict__ rp, const int* __restrict__ col,
                                                 const long long* __restrict__ trp, int* __restrict__ cursor, int* __restrict__ tcol) {
    const long long r = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (r >= n) return;
    for (long long k = rp[r] + (threadIdx.x & 15); k < rp[r + 1]; k += 16) {
        const int j = col[k];
        const int p = atomicAdd(&cursor[j], 1);
        tcol[trp[j] + p] = (int)r;
    }
}
// one wavefront per row: ascending order by a bitonic network in LDS (rows of up to SORT_MAX entries; longer rows are sorted by
// a single lane with insertion - never the case for the reference's stencils, ~275 entries at most on hex meshes)
constexpr int SORT_MAX = 2048;
__global__ __launch_bounds__(256) void k_sort_rows(long long n, const long long* __restrict__ trp, int* __restrict__ tcol) {
    extern __shared__ int sh_sort[];  // 4 waves x SORT_MAX
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + wave;
    if (r >= n) return;
    const long long b = trp[r];
    const int len = (int)(trp[r + 1] - b);
    if (len <= 1) return;
    if (len > SORT_MAX) {
        if (lane == 0)
            for (int i = 1; i < len; i++) {
                const int v = tcol[b + i];
                int q = i - 1;
                while (q >= 0 && tcol[b + q] > v) { tcol[b + q + 1] = tcol[b + q]; q--; }
                tcol[b + q + 1] = v;
            }
        return;
    }
    int* a = sh_sort + wave * SORT_MAX;
    int m = 2;
    while (m < len) m <<= 1;
    for (int i = lane; i < m; i += 64) a[i] = i < len ? tcol[b + i] : 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    for (int k2 = 2; k2 <= m; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int i = lane; i < m; i += 64) {
                const int l = i ^ j2;
                if (l > i) {
                    const int x = a[i], y = a[l];
                    const bool up = (i & k2) == 0;
                    if ((x > y) == up) { a[i] = y; a[l] = x; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    for (int i = lane; i < len; i += 64) tcol[b + i] = a[i];
}

struct DevPattern {  // a connectivity pattern on the device
    long long n = 0, nnz = 0;
    DevBuf<long long> rowptr;
    DevBuf<int> col;
};
// transposed structure (rows sorted ascending) of a row-major device pattern
inline void device_transpose(const DevPattern& P, DevBuf<long long>& trp, DevBuf<int>& tcol, hipStream_t st) {
    const long long n = P.n, nnz = P.nnz;
    DevBuf<int> cnt((size_t)n);
    DAS_HIP(hipMemsetAsync(cnt.p, 0, n * sizeof(int), st));
    hipLaunchKernelGGL(k_tr_count, dim3((unsigned)std::min<long long>((nnz + 255) / 256, 1LL << 22)), dim3(256), 0, st, nnz, P.col.p, cnt.p);
    DAS_HIP(hipGetLastError());
    trp.alloc((size_t)n + 1);
    const long long tot = device_exclusive_scan(n, cnt.p, trp.p, st);
    DAS_CHECK(tot == nnz, DAS_ERR_INTERNAL, "device transpose: entry count mismatch");
    tcol.alloc((size_t)std::max<long long>(1, nnz));
    DAS_HIP(hipMemsetAsync(cnt.p, 0, n * sizeof(int), st));
    hipLaunchKernelGGL(k_tr_fill, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, n, P.rowptr.p, P.col.p, trp.p, cnt.p, tcol.p);
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)((n + 3) / 4)), dim3(256), (size_t)4 * SORT_MAX * sizeof(int), st, n, trp.p, tcol.p);
    DAS_HIP(hipGetLastError());
    DAS_HIP(hipStreamSynchronize(st));
}

// ---- colouring input: nets of every column ---------------------------------------------------------------------------
// netOfRow[r] = position of row r in the kept-row list or -1
__global__ __launch_bounds__(256) void k_net_count(long long n, const long long* __restrict__ trp, const int* __restrict__ tcol, const int* __restrict__ netOfRow,
                                                   int* __restrict__ cnt) {
    const long long j = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (j >= n) return;
    int a = 0;
    for (long long k = trp[j] + (threadIdx.x & 15); k < trp[j + 1]; k += 16) a += netOfRow[tcol[k]] >= 0 ? 1 : 0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_down(a, o, 16);
    if ((threadIdx.x & 15) == 0) cnt[j] = a;
}
// 16 lanes per column, 16 entries per step in ascending row order (an exclusive prefix over the lanes keeps the order): net id
// and the position of the column inside the net's ascending column list (binary search in the row-major row)
__global__ __launch_bounds__(256) void k_net_fill(long long n, const long long* __restrict__ trp, const int* __restrict__ tcol, const int* __restrict__ netOfRow,
                                                  const long long* __restrict__ rp, const int* __restrict__ col, const long long* __restrict__ cptr,
                                                  int* __restrict__ crow, int* __restrict__ cpos) {
    const long long j = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (j >= n) return;
    long long o = cptr[j];
    const long long k0 = trp[j], k1 = trp[j + 1];
    for (long long kb = k0; kb < k1; kb += 16) {
        const long long k = kb + l16;
        int r = 0, net = -1;
        if (k < k1) { r = tcol[k]; net = netOfRow[r]; }
        const int f = net >= 0 ? 1 : 0;
        int incl = f;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { const int v = __shfl_up(incl, d, 16); if (l16 >= d) incl += v; }
        const int total = __shfl(incl, 15, 16);
        if (f) {
            long long lo = rp[r], hi = rp[r + 1] - 1;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (col[mid] < (int)j) lo = mid + 1; else hi = mid;
            }
            const long long dst = o + incl - 1;
            crow[dst] = net;
            cpos[dst] = (int)(lo - rp[r]);
        }
        o += total;
    }
}
// isStart[j] = the net list of column j differs from the one of column j - 1
__global__ __launch_bounds__(256) void k_group_flags(long long n, const long long* __restrict__ cptr, const int* __restrict__ crow, unsigned char* __restrict__ isStart) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (j == 0) { isStart[0] = 1; return; }
    const long long a = cptr[j], len = cptr[j + 1] - a, b = cptr[j - 1];
    bool same = (a - b) == len;
    for (long long q = 0; same && q < len; q++) same = crow[a + q] == crow[b + q];
    isStart[j] = same ? 0 : 1;
}

// out[dst[q] ..] = the columns of row rows[q] of a CSR pattern (one wavefront per selected row)
__global__ __launch_bounds__(256) void k_rows_gather(long long nSel, const long long* __restrict__ rows, const long long* __restrict__ rowptr,
                                                     const int* __restrict__ col, const long long* __restrict__ dst, int* __restrict__ out) {
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= nSel) return;
    const long long b = rowptr[rows[q]], len = rowptr[rows[q] + 1] - b, o = dst[q];
    for (long long k = lane; k < len; k += 64) out[o + k] = col[b + k];
}

}  // namespace das
