// Distance-2 column colouring on the device: the SERIAL first-fit (columns in index order, smallest colour not used by any
// column that shares a kept row) evaluated as a data-flow computation.
//
// Reference: DAColoring::parallelD2Coloring (DAColoring.C:32-784) produces a valid colouring by parallel sweeps with random
// tie-breaks; any valid colouring gives the same Jacobian (validateColoring, DAColoring.C:931-1037, is the contract).  The
// host implementation of this repo (das_jaccon.cpp) is a first-fit; beyond 20 k cells it trades ~20 % more colours for
// tile parallelism.  On the MI355X the serial first-fit itself parallelises: column j only needs the FINAL colours of the
// lower-numbered columns in its neighbourhood, so every column (group) is handed to a wavefront in index order (ticket
// counter) and polls the colours it depends on until they are written - the same "the data is the flag" protocol as the
// preconditioner sweeps (das_bilu.hpp).  Result: exactly the colours of the serial first-fit (410 instead of ~500 at
// 200 k cells, i.e. ~20 % fewer residual passes per Jacobian), in a fraction of the host time.
#pragma once
#include <cstdlib>

#include "das_common.hpp"

namespace das {

constexpr int COLOR_BITWORDS = 64;  // 64 x 64 = 4096 colours per forbidden set (LDS bitmap per wavefront)

struct ColorView {
    long long nGroups;
    const long long* gstart;   // nGroups+1: first column of every group (columns of a group share their kept-row list)
    const long long* cptr;     // n+1: kept rows of a column
    const int* crow;           // kept-row ids
    const long long* krp;      // nKeep+1
    const int* kcol;           // columns of the kept rows
    int* colors;               // n, -1 = not coloured yet
    unsigned* ctrl;            // [0] ticket (one per group), [1] abort / overflow flag
};

// One workgroup per column group (ticket order = column order).  The four wavefronts share the kept rows of the group's
// first column; every wavefront takes two rows at a time and keeps up to ten neighbour-colour gathers in flight per lane
// (the per-group latency is what the dependent chain of the first-fit multiplies).
__global__ __launch_bounds__(256) void k_color_firstfit(ColorView P) {
    __shared__ unsigned sh_ticket;
    __shared__ unsigned long long fb[COLOR_BITWORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (;;) {
        __syncthreads();  // the previous group is done with fb / sh_ticket
        if (threadIdx.x == 0) sh_ticket = atomicAdd(&P.ctrl[0], 1u);
        if (threadIdx.x < COLOR_BITWORDS) fb[threadIdx.x] = 0ull;
        __syncthreads();
        const long long g = sh_ticket;
        if (g >= P.nGroups) return;
        const long long j0 = P.gstart[g], j1 = P.gstart[g + 1];
        const long long q0 = P.cptr[j0], q1 = P.cptr[j0 + 1];
        for (long long q = q0 + 2 * wave; q < q1; q += 8) {
            int jn[10], c[10];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool rowOk = q + h < q1;
                const int r = rowOk ? P.crow[q + h] : 0;
                const long long k0 = P.krp[r], k1 = rowOk ? P.krp[r + 1] : k0;
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    const long long k = k0 + lane + 64 * u;
                    jn[5 * h + u] = k < k1 ? P.kcol[k] : 0x7fffffff;
                }
                // rows longer than 320 entries: the tail goes through the generic loop below
                for (long long k = k0 + lane + 320; k < k1; k += 64) {
                    const int jx = P.kcol[k];
                    if (jx >= j0) continue;
                    int cx = P.colors[jx];
                    while (cx < 0) cx = __hip_atomic_load(&P.colors[jx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cx < 64 * COLOR_BITWORDS) atomicOr(&fb[cx >> 6], 1ull << (cx & 63));
                    else __hip_atomic_store(&P.ctrl[1], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < 10; u++) c[u] = jn[u] < j0 ? P.colors[jn[u]] : 0;  // (may be a stale -1 from this CU's L1)
#pragma unroll
            for (int u = 0; u < 10; u++) {
                if (jn[u] >= j0) continue;  // not coloured yet in the serial order, a member of this group, or padding
                int cx = c[u];
                unsigned spins = 0;
                while (cx < 0) {
                    __builtin_amdgcn_s_sleep(4);  // a waiting lane must not flood the memory system (polling-cost)
                    cx = __hip_atomic_load(&P.colors[jn[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cx < 0 && (++spins & 1023u) == 0u) {
                        if (__hip_atomic_load(&P.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 24)) {
                            __hip_atomic_store(&P.ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            cx = 0;
                        }
                    }
                }
                if (cx >= 64 * COLOR_BITWORDS) { __hip_atomic_store(&P.ctrl[1], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); cx = 64 * COLOR_BITWORDS - 1; }
                atomicOr(&fb[cx >> 6], 1ull << (cx & 63));
            }
        }
        __syncthreads();
        if (wave == 0) {
            // the members of the group take the smallest free colours one after the other
            unsigned long long mine = fb[lane];
            for (long long j = j0; j < j1; j++) {
                const unsigned long long w = ~mine;
                int best = w ? (lane * 64 + __builtin_ctzll(w)) : (1 << 30);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
                if (best >= 64 * COLOR_BITWORDS) { best = 64 * COLOR_BITWORDS - 1; if (lane == 0) __hip_atomic_store(&P.ctrl[1], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                if (lane == (best >> 6)) mine |= 1ull << (best & 63);
                if (lane == 0) __hip_atomic_store(&P.colors[j], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// colours of the serial first-fit on the device.  keep/cptr/crow: the kept (non-dominated) rows and the CSC over them
// (das_jaccon.cpp); returns false if the device path could not be used (more than 4096 colours, timeout): the caller then
// falls back to... nothing - the host algorithm is run instead, loudly.
inline bool color_firstfit_device(long long n, const std::vector<long long>& keep, const std::vector<long long>& cptr, const uvector<int>& crow,
                                  const std::vector<long long>& rowptr, const uvector<int>& col, std::vector<int>& colors, hipStream_t st) {
    const long long nKeep = (long long)keep.size();
    const bool dbg = getenv("DAS_DEBUG_TIMING") != nullptr;
    double tq = wall_seconds();
    auto lap = [&](const char* what) { if (dbg) { double t2 = wall_seconds(); fprintf(stderr, "[dafoam_amd]     device colouring: %s %.2f s\n", what, t2 - tq); tq = t2; } };
    // compact pattern of the kept rows
    std::vector<long long> krp(nKeep + 1, 0);
    for (long long q = 0; q < nKeep; q++) krp[q + 1] = krp[q] + (rowptr[keep[q] + 1] - rowptr[keep[q]]);
    uvector<int> kcol(krp[nKeep]);
#pragma omp parallel for schedule(static)
    for (long long q = 0; q < nKeep; q++) std::copy(col.begin() + rowptr[keep[q]], col.begin() + rowptr[keep[q] + 1], kcol.begin() + krp[q]);
    // crow holds row ids of the full pattern: translate to kept-row positions
    std::vector<int> posOfRow;
    {
        long long nrows = (long long)rowptr.size() - 1;
        posOfRow.assign(nrows, -1);
        for (long long q = 0; q < nKeep; q++) posOfRow[keep[q]] = (int)q;
    }
    uvector<int> crowK(crow.size());
#pragma omp parallel for schedule(static)
    for (long long q = 0; q < (long long)crow.size(); q++) crowK[q] = posOfRow[crow[q]];
    // groups: maximal runs of consecutive columns with identical kept-row lists (the xyz components of a cell's U)
    std::vector<long long> gstart;
    {
        std::vector<unsigned char> isStart(n, 1);
#pragma omp parallel for schedule(static)
        for (long long j = 1; j < n; j++) {
            const long long len = cptr[j + 1] - cptr[j];
            isStart[j] = !(cptr[j] - cptr[j - 1] == len && std::equal(crow.begin() + cptr[j], crow.begin() + cptr[j + 1], crow.begin() + cptr[j - 1]));
        }
        gstart.reserve(n);
        for (long long j = 0; j < n; j++) if (isStart[j]) gstart.push_back(j);
    }
    lap("host preparation");
    const long long nGroups = (long long)gstart.size();
    gstart.push_back(n);
    DevBuf<long long> d_gstart, d_cptr, d_krp;
    DevBuf<int> d_crow, d_kcol, d_colors(n);
    DevBuf<unsigned> d_ctrl(2);
    d_gstart.upload(gstart); d_cptr.upload(cptr); d_krp.upload(krp);
    d_crow.upload(crowK.data(), crowK.size()); d_kcol.upload(kcol.data(), kcol.size());
    DAS_HIP(hipMemsetAsync(d_colors.p, 0xff, n * sizeof(int), st));
    DAS_HIP(hipMemsetAsync(d_ctrl.p, 0, 2 * sizeof(unsigned), st));
    DAS_HIP(hipStreamSynchronize(st));
    lap("upload");
    ColorView V{nGroups, d_gstart.p, d_cptr.p, d_crow.p, d_krp.p, d_kcol.p, d_colors.p, d_ctrl.p};
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // workgroups in flight = the window of consecutive column groups being coloured.  Neighbouring groups conflict, so the
    // parallelism inside the window comes from the rows / planes of cells it spans: measured, the kernel time falls like
    // 1 / window up to the residency limit (200 k cells: 5.8 s with 64 workgroups, 0.8 s with 1024; profiles/README.md)
    long long wgs = (long long)cus * 8;
    if (const char* e = getenv("DAS_COLOR_WGS")) wgs = std::max(1, atoi(e));
    const int grid = (int)std::min<long long>(wgs, nGroups + 1);
    hipLaunchKernelGGL(k_color_firstfit, dim3(grid), dim3(256), 0, st, V);
    DAS_HIP(hipGetLastError());
    unsigned ctrl[2] = {0, 0};
    DAS_HIP(hipMemcpyAsync(ctrl, d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost, st));
    DAS_HIP(hipStreamSynchronize(st));
    lap("kernel");
    if (ctrl[1] != 0u) return false;
    colors.resize(n);
    d_colors.download(colors.data(), n);
    return true;
}

}  // namespace das
