// Dev tool (GPU): latency of one producer -> consumer "hop" through global memory between two workgroups, the unit cost of the
// sync-free triangular sweeps (csrc/das_bilu.hpp: ~925 dependent hops per sweep at 2 M cells).  Ping-pong between workgroup A and B:
// A publishes i, B polls until it sees i and publishes i in its own word, A polls that, ...  Variants:
//   store sc1 (agent scope) + poll with agent-scope loads            - what k_bilu_sweep does today
//   store sc1 + poll with a workgroup-scope L2 atomic (fetch_or 0)   - stays in the XCD's L2 when both groups sit on one XCD
// for pairs on the SAME XCD and on DIFFERENT XCDs (XCC_ID read at run time; the grid is sized so that both kinds occur).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hop tools/gpu/hop_latency.hip && /tmp/hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned xcc() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }
template <int MODE>
__global__ void k_pingpong(unsigned long long* flags, unsigned* xcdOf, int iters, long long* cycles, int pairs, int shift) {
    // blocks [0, pairs) = side A of pair p, blocks [pairs, 2 pairs) = side B of pair (b - pairs - shift) mod pairs: with pairs a multiple of
    // 8 and the round-robin placement of workgroups, shift 0 puts both sides on one XCD, shift 1 on neighbouring XCDs (checked via XCC_ID)
    const int side = blockIdx.x >= pairs;
    const int p = side ? (int)((blockIdx.x - pairs - shift + pairs) % pairs) : (int)blockIdx.x;
    const int other = side ? p : pairs + (p + shift) % pairs;
    if (threadIdx.x == 0) xcdOf[blockIdx.x] = xcc();
    if (threadIdx.x != 0) return;
    unsigned long long* mine = flags + (size_t)blockIdx.x * 32;
    unsigned long long* theirs = flags + (size_t)other * 32;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= iters; i++) {
        if (side == 0) __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        for (;;) {
            unsigned long long v;
            if (MODE == 0) v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v = __hip_atomic_fetch_or(theirs, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v >= (unsigned long long)i) break;
            if (++spins > (1u << 21)) {  // never seen (e.g. a stale line in this XCD's L2 with the L2-scope poll across XCDs): give up, release the partner
                __hip_atomic_store(mine, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (side == 0) cycles[p] = -1;
                return;
            }
        }
        if (side == 1) __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (side == 0) cycles[p] = wall_clock64() - t0;
}
int main() {
    const int pairs = 64, iters = 2000;
    unsigned long long* flags; unsigned* xcd; long long* cyc;
    hipMalloc(&flags, (size_t)pairs * 2 * 32 * 8); hipMalloc(&xcd, pairs * 2 * 4); hipMalloc(&cyc, pairs * 8);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);  // kHz
    for (int mode = 0; mode < 2; mode++) {
        double same = 0, diff = 0; int ns = 0, nd = 0;
        for (int shift = 0; shift < 2; shift++) {
            hipMemset(flags, 0, (size_t)pairs * 2 * 32 * 8);
            if (mode == 0) hipLaunchKernelGGL(k_pingpong<0>, dim3(pairs * 2), dim3(64), 0, 0, flags, xcd, iters, cyc, pairs, shift);
            else hipLaunchKernelGGL(k_pingpong<1>, dim3(pairs * 2), dim3(64), 0, 0, flags, xcd, iters, cyc, pairs, shift);
            hipDeviceSynchronize();
            std::vector<unsigned> hx(pairs * 2); std::vector<long long> hc(pairs);
            hipMemcpy(hx.data(), xcd, pairs * 2 * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), cyc, pairs * 8, hipMemcpyDeviceToHost);
            for (int p = 0; p < pairs; p++) {
                if (hc[p] < 0) { printf("  pair %d (XCD %u / %u): the poller never saw the store\n", p, hx[p], hx[pairs + (p + shift) % pairs]); continue; }
                const double us = (double)hc[p] / (double)rate * 1e3 / (2.0 * iters);
                if (hx[p] == hx[pairs + (p + shift) % pairs]) { same += us; ns++; } else { diff += us; nd++; }
            }
        }
        printf("%s: one hop (store -> seen by the poller) same XCD %.3f us (%d pairs), different XCDs %.3f us (%d pairs)\n",
               mode == 0 ? "agent-scope load poll     " : "workgroup-scope L2 atomic", ns ? same / ns : 0.0, ns, nd ? diff / nd : 0.0, nd);
    }
    return 0;
}
