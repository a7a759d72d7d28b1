// Global node-block ILU(0) preconditioner with sync-free (data-flow) triangular sweeps - gfx950 only.
//
// What it replaces: the reference's PC stack ASM + ILU(pcFillLevel) (DALinearEqn.C:199-299).  With ONE sub-domain per
// GPU (the reference's layout is one ASM sub-domain per MPI rank) the additive-Schwarz level disappears inside a rank
// and the preconditioner is a single incomplete factorisation of dRdWTPC restricted to the owned unknowns.
//
// MI355X design:
//   * NODES: the unknowns are grouped cell by cell (adjStateOrdering "cell", DAIndex.C:602-651) into dense nodes of
//     BILU_NB = 8 slots (an interior hex cell of DASimpleFoam+SA - U, p, nuTilda and the phi of its three owned faces -
//     is exactly one node; other cells are split / packed, empty slots are identity rows).  The factorisation is the
//     block ILU(0) on the node graph: node I is coupled to every node whose cells lie within the stencil reach of the
//     PC connectivity (reduceStateResConLevel, DASolver.C:576-705).  It contains every entry of the scalar ILU(0) plus
//     the fill inside the 8x8 blocks; there are no per-entry indices (8 B per factor entry instead of 12) and the
//     dependent chain of a triangular solve shrinks from ~8 (nx+2ny+3nz) scalar levels to nx+2ny+3nz node levels.
//   * ORDER: nodes are renumbered by level set of the lower-triangular dependency graph (a topological order that keeps
//     the relative order of all coupled nodes, so the factorisation equals the natural-order one up to summation
//     order).  Consecutive nodes are independent.
//   * FACTORISATION on the device: one launch per level, one wavefront per node row, 8x8 products through wave
//     shuffles, block inverse by Gauss-Jordan with row pivoting and a non-zero pivot shift (PCFactorSetShiftType
//     NONZERO analogue, DALinearEqn.C:270-272).
//   * SWEEPS: one wavefront per node, no level barriers and no flags: the solution vector itself is the flag (all
//     entries start as a NaN sentinel; a node polls the 8 values of each dependency with agent-scope (sc1) loads until
//     they are written, then publishes its own 8 values with sc1 stores - "the data IS the flag",
//     cdna_hip_programming.md Guideline 16 R2).  Work is handed out in processing order through a device-side ticket
//     counter, so a waiting wave only ever waits for nodes that are already running: no dependence on dispatch order
//     or residency.  Factor blocks are streamed exactly once per sweep, packed per 8-block pass so that every
//     16-byte load instruction of a wave reads one contiguous segment.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <numeric>

#include <omp.h>

#include "das_common.hpp"
#include "das_jaccon.hpp"

namespace das {

constexpr int BILU_NB = 8;
constexpr int BILU_NB2 = 64;
constexpr unsigned long long BILU_SENTINEL = 0xFFF7A5A5FFF7A5A5ull;  // a NaN no arithmetic produces
#ifndef BILU_WG
#define BILU_WG 512  // threads per workgroup of the sweeps (8 wavefronts)
#endif
#ifndef BILU_OCC
#define BILU_OCC 4   // wavefronts per SIMD the sweeps are compiled for (register budget 512 / BILU_OCC)
#endif
#ifndef BILU_XCD_TICKETS
#define BILU_XCD_TICKETS 1  // allow one ticket counter per XCD (see k_bilu_sweep) where the host finds it safe; 0: always one device-wide counter
#endif
constexpr int BILU_CTRL_STRIDE = 32;                       // unsigneds per 128-byte line
constexpr int BILU_CTRL_ABORT = 16 * BILU_CTRL_STRIDE;     // lines 0-7: forward-sweep counters, 8-15: backward, 16: abort flag
constexpr int BILU_CTRL_SIZE = 17 * BILU_CTRL_STRIDE;
#ifndef BILU_SPIN_LIMIT
#define BILU_SPIN_LIMIT (1u << 22)
#endif

struct BiluView {
    int nNodes;
    const int* nodeUnk;          // nNodes*8: global state index of a slot or -1 (processing order)
    const int* nodeOut;          // nNodes*8: where the slot's solution is written (= nodeUnk; -1 for the overlap copies of a multi-block factorisation)
    const long long* ptr[2];     // [0] L rows by p, [1] U rows by q = nNodes-1-p
    const int* col[2];           // node positions p of the dependencies
    const double* val[2];
    const float* valf[2];
    const double* invD;          // 64 per node, row-major
    double* y;                   // forward-sweep result (also the flags of the forward sweep)
    double* z;                   // backward-sweep result
    unsigned* ctrl;              // ticket counters and abort flag, one 128-byte line each (BILU_CTRL_* below)
};

// ---------------------------------------------------------------------------------------------------
// device kernels
// ---------------------------------------------------------------------------------------------------
__global__ void k_bilu_reset(long long nslots, double* y, double* z, unsigned* ctrl) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        reinterpret_cast<unsigned long long*>(y)[i] = BILU_SENTINEL;
        reinterpret_cast<unsigned long long*>(z)[i] = BILU_SENTINEL;
    }
    if (i < 16) ctrl[i * BILU_CTRL_STRIDE] = 0u;
}

__device__ __forceinline__ void bilu_store_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long bilu_load_sc1(const double* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class VT>
__device__ __forceinline__ void bilu_load_pair(const VT* p, double& a, double& b);
template <>
__device__ __forceinline__ void bilu_load_pair<double>(const double* p, double& a, double& b) {
    const double2 t = *reinterpret_cast<const double2*>(p);
    a = t.x; b = t.y;
}
template <>
__device__ __forceinline__ void bilu_load_pair<float>(const float* p, double& a, double& b) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    a = (double)t.x; b = (double)t.y;
}

// One triangular sweep.  UPPER = false: y_p = b_p - sum_{J<p} L_pJ y_J; UPPER = true: z_p = invD_p (y_p - sum_{J>p} U_pJ z_J),
// out[state] = z.  Lane (g, k) = (lane / 8, lane % 8) owns COLUMN k of the g-th block of a pass of 8 blocks: it polls exactly
// the one solution value it multiplies with (x_k of dependency g), keeps 8 row accumulators over all passes, and one
// reduce-scatter per node (10 exchanges) leaves the row sums in the lane groups.
// (Measured and dropped: a lean variant without the prefetch of the next pass - <= 64 VGPRs, 4 workgroups per CU - and
// several nodes per wave per ticket: both slower at 200 k and at 2 M cells.)
template <class VT, bool UPPER>
__global__ __launch_bounds__(BILU_WG, BILU_OCC) void k_bilu_sweep(BiluView P, const double* __restrict__ b, double* __restrict__ out, int sleepReps,
                                                                   int perXcd) {
    __shared__ unsigned sh_chunk[2];
    constexpr int WAVES = BILU_WG / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, k = lane & 7;
    const long long* __restrict__ ptr = P.ptr[UPPER ? 1 : 0];
    const int* __restrict__ col = P.col[UPPER ? 1 : 0];
    const VT* __restrict__ val = reinterpret_cast<const VT*>(sizeof(VT) == 4 ? (const void*)P.valf[UPPER ? 1 : 0] : (const void*)P.val[UPPER ? 1 : 0]);
    double* xs = UPPER ? P.z : P.y;
    // Tickets per XCD (perXcd != 0): a device-wide counter is one address all 8 XCDs increment - an agent-scope atomic executes
    // at the memory side, ~13 ns apiece, and 290 k tickets per sweep at 2 M cells made the counter the pace of the whole sweep.
    // Here XCD x (XCC_ID hardware register) draws local tickets i from its own counter and solves global ticket 8 i + x.  The
    // counter of an XCD is only ever touched by workgroups of that XCD, and gfx9 executes global atomics in the (per-XCD) L2:
    // a workgroup-scope atomic is therefore coherent among exactly the workgroups that use it and never leaves that L2.
    // Still deadlock-free: the lowest unfinished ticket belongs to some XCD, whose workgroups draw their tickets in
    // increasing order and wait only for lower ones.  COMPLETE only if every one of the 8 XCDs runs at least one workgroup
    // of the launch: the host enables the mode only after bilu_xcd_probe() has seen XCC ids 0..7 on this device AND for
    // grids of >= 8 workgroups per XCD (workgroups are dealt round-robin to the XCDs); every other launch - small meshes,
    // partitioned (CPX/DPX/QPX) devices, other parts - takes the device-wide counter (the 2-workgroup
    // launches of the 1- and 2-cell test meshes used to land on two arbitrary XCDs and solve nothing when neither was XCD 0).
    const unsigned xcd = perXcd ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0u;  // HW_REG_XCC_ID[3:0]
    unsigned* ctr = &P.ctrl[((UPPER ? 8 : 0) + xcd) * BILU_CTRL_STRIDE];
    for (unsigned it = 0;; it++) {
        if (threadIdx.x == 0)
            sh_chunk[it & 1] = perXcd ? __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * 8u + xcd
                                      : __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const long long q0 = (long long)sh_chunk[it & 1] * WAVES;
        if (q0 >= P.nNodes) return;
        {
            const long long q = q0 + wave;
            if (q >= P.nNodes) continue;
            const long long e0 = ptr[q];
            const int nE = (int)(ptr[q + 1] - e0);
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            // software pipeline over the passes: the factor blocks of pass i+1 are requested before pass i spins on its
            // dependencies (the block loads never depend on the solution)
            double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int c = 0, cn = 0;
            auto load_pass = [&](int a0, double (&vv)[8], int& cc) {
                const int nb = min(8, nE - a0);
                if (g < nb) {
                    cc = col[e0 + a0 + g];
                    const VT* base = val + (e0 + a0) * BILU_NB2;
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) bilu_load_pair<VT>(base + ((qq * nb + g) * 8 + k) * 2, vv[2 * qq], vv[2 * qq + 1]);
                }
            };
            if (nE > 0) load_pass(0, v, c);
            // what the end of the node needs besides the row sums does not depend on the solution: requested here, so that the
            // (dependent) loads are not on the chain dependency ready -> row sums -> publish
            const long long p = UPPER ? (long long)P.nNodes - 1 - q : q;
            int gi;
            double rhs, dinv = 0.0;
            if (!UPPER) {
                gi = P.nodeUnk[p * BILU_NB + g];
                rhs = gi >= 0 ? b[gi] : 0.0;
            } else {
                gi = P.nodeOut[p * BILU_NB + k];
                rhs = P.y[p * BILU_NB + g];
                dinv = P.invD[p * BILU_NB2 + k * 8 + g];
            }
            for (int a0 = 0; a0 < nE; a0 += 8) {
                const bool act = g < min(8, nE - a0);
                if (a0 + 8 < nE) load_pass(a0 + 8, vn, cn);
                const double* xp = xs + (long long)c * BILU_NB + k;
                unsigned long long xb = 0ull;
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
                    if (act) { xb = bilu_load_sc1(xp); ok = xb != BILU_SENTINEL; }
                    if (__all(ok)) break;
                    for (int w = 0; w < sleepReps; w++) __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023u) == 0u) {  // bounded spin: a stuck sweep sets the abort flag instead of hanging the GPU
                        const unsigned ab = __hip_atomic_load(&P.ctrl[BILU_CTRL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ab != 0u || spins >= BILU_SPIN_LIMIT) {
                            if (lane == 0) __hip_atomic_store(&P.ctrl[BILU_CTRL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            return;
                        }
                    }
                }
                if (act) {
                    const double xk = __longlong_as_double((long long)xb);
#pragma unroll
                    for (int r = 0; r < 8; r++) acc[r] += v[r] * xk;
                }
#pragma unroll
                for (int r = 0; r < 8; r++) v[r] = vn[r];
                c = cn;
            }
            // reduce-scatter over the 8 lane groups: afterwards group g holds (per column lane) the partial sums of row g
            double a4[4], a2[2], a1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double snd = (g & 4) ? acc[i] : acc[i + 4], keep = (g & 4) ? acc[i + 4] : acc[i];
                a4[i] = keep + __shfl_xor(snd, 32, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const double snd = (g & 2) ? a4[i] : a4[i + 2], keep = (g & 2) ? a4[i + 2] : a4[i];
                a2[i] = keep + __shfl_xor(snd, 16, 64);
            }
            {
                const double snd = (g & 1) ? a2[0] : a2[1], keep = (g & 1) ? a2[1] : a2[0];
                a1 = keep + __shfl_xor(snd, 8, 64);
            }
            a1 += __shfl_xor(a1, 1, 64);
            a1 += __shfl_xor(a1, 2, 64);
            a1 += __shfl_xor(a1, 4, 64);  // every lane of group g: the full sum of row g
            if (!UPPER) {
                if (k == 0) bilu_store_sc1(&P.y[p * BILU_NB + g], rhs - a1);
            } else {
                double w = dinv * (rhs - a1);  // row k of invD times t, summed over the groups
                w += __shfl_xor(w, 8, 64);
                w += __shfl_xor(w, 16, 64);
                w += __shfl_xor(w, 32, 64);
                if (g == 0) {
                    bilu_store_sc1(&P.z[p * BILU_NB + k], w);
                    if (gi >= 0) out[gi] = w;
                }
            }
        }
    }
}

// (Round 3, measured and dropped - profiles/r03f_*: a second design with one ticket per WAVE drawn two nodes ahead and the row
// extent / first pass of the next node requested while the current one waits for its dependencies, no workgroup barrier.
// Parity-clean, but slower everywhere: 1.40 -> 1.97 ms at 200 k cells, 4.56 -> 7.15 ms on the 200 k-cell NACA0012 O-grid,
// 6.36 -> 10.3 ms at 2 M cells.  Eight times the tickets means eight times the atomics on the same eight addresses, and a
// contended L2 atomic retires every ~17 ns: 290 k tickets per XCD and sweep at 2 M cells = 5 ms.  Together with the fp32
// result (half the factor bytes, same time) this fixes the picture of the sweeps: neither bandwidth nor the per-node load
// chain bounds them, the ticket rate and the dependent hops do; 8 nodes per ticket is the measured optimum.)
// ---- the same sweep for S right-hand sides at once (block GMRES, das_block.hpp) ------------------------------------
// The sweeps are bound by the ticket rate and the dependent hops, not by bytes (see above), so S systems through ONE sweep cost
// little more than one: the ticket, the row extent, the factor blocks and the polling round trip of a node are shared, only the
// multiply-accumulate, the reduce-scatter and the published values are per system.  Work vectors ym / zm hold S values per
// slot, [(node * 8 + slot) * S + r], so that a lane polls the S values of "its" x_k with one 16- or 32-byte load; b and out
// are column-major n x S (ld = leading dimension).
template <class VT, bool UPPER, int S>
__global__ __launch_bounds__(BILU_WG, BILU_OCC) void k_bilu_sweep_m(BiluView P, double* __restrict__ ym, double* __restrict__ zm, const double* __restrict__ b,
                                                                     double* __restrict__ out, long long ld, int sleepReps, int perXcd) {
    __shared__ unsigned sh_chunk[2];
    constexpr int WAVES = BILU_WG / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 3, k = lane & 7;
    const long long* __restrict__ ptr = P.ptr[UPPER ? 1 : 0];
    const int* __restrict__ col = P.col[UPPER ? 1 : 0];
    const VT* __restrict__ val = reinterpret_cast<const VT*>(sizeof(VT) == 4 ? (const void*)P.valf[UPPER ? 1 : 0] : (const void*)P.val[UPPER ? 1 : 0]);
    double* xs = UPPER ? zm : ym;
    const unsigned xcd = perXcd ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0u;
    unsigned* ctr = &P.ctrl[((UPPER ? 8 : 0) + xcd) * BILU_CTRL_STRIDE];
    for (unsigned it = 0;; it++) {
        if (threadIdx.x == 0)
            sh_chunk[it & 1] = perXcd ? __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) * 8u + xcd
                                      : __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const long long q0 = (long long)sh_chunk[it & 1] * WAVES;
        if (q0 >= P.nNodes) return;
        const long long q = q0 + wave;
        if (q >= P.nNodes) continue;
        const long long e0 = ptr[q];
        const int nE = (int)(ptr[q + 1] - e0);
        double acc[S][8];
#pragma unroll
        for (int r = 0; r < S; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) acc[r][i] = 0.0;
        double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, vn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int c = 0, cn = 0;
        auto load_pass = [&](int a0, double (&vv)[8], int& cc) {
            const int nb = min(8, nE - a0);
            if (g < nb) {
                cc = col[e0 + a0 + g];
                const VT* base = val + (e0 + a0) * BILU_NB2;
#pragma unroll
                for (int qq = 0; qq < 4; qq++) bilu_load_pair<VT>(base + ((qq * nb + g) * 8 + k) * 2, vv[2 * qq], vv[2 * qq + 1]);
            }
        };
        if (nE > 0) load_pass(0, v, c);
        const long long p = UPPER ? (long long)P.nNodes - 1 - q : q;
        int gi;
        double rhs[S], dinv = 0.0;
        if (!UPPER) {
            gi = P.nodeUnk[p * BILU_NB + g];
#pragma unroll
            for (int r = 0; r < S; r++) rhs[r] = gi >= 0 ? b[gi + r * ld] : 0.0;
        } else {
            gi = P.nodeOut[p * BILU_NB + k];
#pragma unroll
            for (int r = 0; r < S; r++) rhs[r] = ym[(p * BILU_NB + g) * S + r];
            dinv = P.invD[p * BILU_NB2 + k * 8 + g];
        }
        for (int a0 = 0; a0 < nE; a0 += 8) {
            const bool act = g < min(8, nE - a0);
            if (a0 + 8 < nE) load_pass(a0 + 8, vn, cn);
            const double* xp = xs + ((long long)c * BILU_NB + k) * S;
            unsigned long long xb[S];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                if (act) {
#pragma unroll
                    for (int r = 0; r < S; r++) { xb[r] = bilu_load_sc1(xp + r); ok = ok && xb[r] != BILU_SENTINEL; }
                }
                if (__all(ok)) break;
                for (int w = 0; w < sleepReps; w++) __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0u) {
                    const unsigned ab = __hip_atomic_load(&P.ctrl[BILU_CTRL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ab != 0u || spins >= BILU_SPIN_LIMIT) {
                        if (lane == 0) __hip_atomic_store(&P.ctrl[BILU_CTRL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        return;
                    }
                }
            }
            if (act) {
#pragma unroll
                for (int r = 0; r < S; r++) {
                    const double xk = __longlong_as_double((long long)xb[r]);
#pragma unroll
                    for (int i = 0; i < 8; i++) acc[r][i] += v[i] * xk;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = vn[i];
            c = cn;
        }
#pragma unroll
        for (int r = 0; r < S; r++) {
            double a4[4], a2[2], a1;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const double snd = (g & 4) ? acc[r][i] : acc[r][i + 4], keep = (g & 4) ? acc[r][i + 4] : acc[r][i];
                a4[i] = keep + __shfl_xor(snd, 32, 64);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const double snd = (g & 2) ? a4[i] : a4[i + 2], keep = (g & 2) ? a4[i + 2] : a4[i];
                a2[i] = keep + __shfl_xor(snd, 16, 64);
            }
            {
                const double snd = (g & 1) ? a2[0] : a2[1], keep = (g & 1) ? a2[1] : a2[0];
                a1 = keep + __shfl_xor(snd, 8, 64);
            }
            a1 += __shfl_xor(a1, 1, 64);
            a1 += __shfl_xor(a1, 2, 64);
            a1 += __shfl_xor(a1, 4, 64);
            if (!UPPER) {
                if (k == 0) bilu_store_sc1(&ym[(p * BILU_NB + g) * S + r], rhs[r] - a1);
            } else {
                double w = dinv * (rhs[r] - a1);
                w += __shfl_xor(w, 8, 64);
                w += __shfl_xor(w, 16, 64);
                w += __shfl_xor(w, 32, 64);
                if (g == 0) {
                    bilu_store_sc1(&zm[(p * BILU_NB + k) * S + r], w);
                    if (gi >= 0) out[gi + r * ld] = w;
                }
            }
        }
    }
}

// scalar CSR (rows = states, 16 lanes per row) -> dense node blocks (row-major 8x8 per block entry)
__global__ __launch_bounds__(256) void k_bilu_scatter(long long n, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                      const double* __restrict__ v, const int* __restrict__ unkNode,
                                                      const unsigned char* __restrict__ unkSlot, const long long* __restrict__ bptr,
                                                      const int* __restrict__ bcol, const unsigned char* __restrict__ late, double* __restrict__ bval,
                                                      unsigned long long* dropped, int transpose, double diagScale, long long shiftExLo,
                                                      long long shiftExHi, long long shiftEnd) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l16 = threadIdx.x & 15;
    if (row >= n) return;
    const int I0 = unkNode[row];
    if (I0 < 0) return;
    const int r0 = unkSlot[row];
    for (long long k = rp[row] + l16; k < rp[row + 1]; k += 16) {
        const int j = ci[k];
        const int J0 = unkNode[j];
        if (J0 < 0) continue;  // not owned by this rank (block-Jacobi across ranks)
        // transpose: the factorisation of the TRANSPOSED matrix (the forward system dR/dW of the Newton primal): entry (i,j)
        // goes to block (J,I), element (c,r); the node pattern is symmetric
        const int I = transpose ? J0 : I0, J = transpose ? I0 : J0;
        const int r = transpose ? unkSlot[j] : r0, c = transpose ? r0 : unkSlot[j];
        long long lo = bptr[I], hi = bptr[I + 1] - 1, e = -1;
        while (lo <= hi) {
            const long long mid = (lo + hi) >> 1;
            const int cm = bcol[mid];
            if (cm == J) { e = mid; break; }
            if (cm < J) lo = mid + 1; else hi = mid - 1;
        }
        if (e < 0) { if (!(late[I] && late[J])) atomicAdd(dropped, 1ull); continue; }  // late-late couplings are dropped by design
        // diagScale = 1 + 1/tau: pseudo-transient shift, on the rows below shiftEnd outside [shiftExLo, shiftExHi) (the Newton primal
        // leaves the pressure and flux rows unshifted, see run_newton_primal)
        const bool shifted = row < shiftEnd && !(row >= shiftExLo && row < shiftExHi);
        bval[e * BILU_NB2 + r * 8 + c] = (j == row && shifted) ? v[k] * diagScale : v[k];
    }
}
__global__ void k_bilu_pad_diag(int nNodes, const int* __restrict__ nodeUnk, const long long* __restrict__ bdiag, double* __restrict__ bval) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nNodes * BILU_NB) return;
    if (nodeUnk[i] < 0) { const int k = (int)(i & 7); bval[bdiag[i >> 3] * BILU_NB2 + k * 8 + k] = 1.0; }
}

// 8x8 inverse of the block held one element per lane (lane = r*8+c): Gauss-Jordan with row pivoting; a pivot below
// `tiny` is replaced by +-shift (non-zero shift).  Returns the inverse element of this lane; *nshift counts shifts.
__device__ __forceinline__ double bilu_inverse8(double a, int lane, int* nshift) {
    const int r = lane >> 3, c = lane & 7;
    double b = (r == c) ? 1.0 : 0.0;
    for (int k = 0; k < 8; k++) {
        int pr = k;
        double best = -1.0;
        for (int rr = k; rr < 8; rr++) {
            const double t = fabs(__shfl(a, rr * 8 + k, 64));
            if (t > best) { best = t; pr = rr; }
        }
        // swap rows k and pr
        const int src = (r == k) ? pr * 8 + c : (r == pr ? k * 8 + c : lane);
        a = __shfl(a, src, 64);
        b = __shfl(b, src, 64);
        double pv = __shfl(a, k * 8 + k, 64);
        if (!(fabs(pv) > 1e-300)) {  // MAT_SHIFT_NONZERO analogue: the pivot is replaced, the elimination continues
            pv = (pv < 0.0 ? -1.0 : 1.0) * 1e-12;
            if (lane == k * 9) a = pv;
            if (lane == 0) (*nshift)++;
        }
        const double rk = __shfl(a, k * 8 + c, 64) / pv;
        const double rkb = __shfl(b, k * 8 + c, 64) / pv;
        const double f = __shfl(a, r * 8 + k, 64);
        if (r == k) { a = rk; b = rkb; }
        else { a -= f * rk; b -= f * rkb; }
    }
    return b;
}

// numeric block ILU(0) of the rows [node0, node1) of one level (all their dependencies belong to earlier launches);
// one wavefront per row, lane = r*8+c holds element (r,c) of the current block
__global__ __launch_bounds__(256) void k_bilu_factor(int node0, int node1, const long long* __restrict__ bptr, const long long* __restrict__ bdiag,
                                                     const int* __restrict__ bcol, double* bval, double* invD, int* nshift) {
    const int lane = threadIdx.x & 63;
    const int p = node0 + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= node1) return;
    const int r = lane >> 3, c = lane & 7;
    const long long rb = bptr[p], rd = bdiag[p], re = bptr[p + 1];
    for (long long e = rb; e < rd; e++) {
        const int J = bcol[e];
        const double a = bval[e * BILU_NB2 + lane];
        double Lv = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) Lv += __shfl(a, r * 8 + k, 64) * invD[(long long)J * BILU_NB2 + k * 8 + c];
        bval[e * BILU_NB2 + lane] = Lv;
        double Lrow[8];
#pragma unroll
        for (int k = 0; k < 8; k++) Lrow[k] = __shfl(Lv, r * 8 + k, 64);
        const long long je = bptr[J + 1];
        long long from = e + 1;
        for (long long f = bdiag[J] + 1; f < je; f++) {
            const int M = bcol[f];
            // both lists are ascending: continue the search behind the previous hit
            long long lo = from, hi = re - 1, pos = -1;
            while (lo <= hi) {
                const long long mid = (lo + hi) >> 1;
                const int cm = bcol[mid];
                if (cm == M) { pos = mid; break; }
                if (cm < M) lo = mid + 1; else hi = mid - 1;
            }
            if (pos < 0) { from = lo; continue; }
            from = pos + 1;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) s += Lrow[k] * bval[f * BILU_NB2 + k * 8 + c];
            bval[pos * BILU_NB2 + lane] -= s;
        }
    }
    int ns = 0;
    const double inv = bilu_inverse8(bval[rd * BILU_NB2 + lane], lane, &ns);
    invD[(long long)p * BILU_NB2 + lane] = inv;
    if (lane == 0 && ns) atomicAdd(nshift, ns);
}

// pack the factor into the two sweep streams: per row the blocks in passes of <= 8, inside a pass [qq][g][k][2]
// (row r = 2 qq + rr of column k), so that the qq-th 16-byte load of all lanes of a wave reads one contiguous segment
template <class VT>
__global__ __launch_bounds__(256) void k_bilu_pack(int nNodes, const long long* __restrict__ bptr, const long long* __restrict__ bdiag,
                                                   const int* __restrict__ bcol, const double* __restrict__ bval, const long long* __restrict__ Lptr,
                                                   const long long* __restrict__ Uptr, int* __restrict__ Lcol, int* __restrict__ Ucol,
                                                   VT* __restrict__ Lval, VT* __restrict__ Uval) {
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= nNodes) return;
    const int r = lane >> 3, k = lane & 7;
    for (int t = 0; t < 2; t++) {
        const long long s0 = t == 0 ? bptr[p] : bdiag[p] + 1;
        const long long s1 = t == 0 ? bdiag[p] : bptr[p + 1];
        const int nE = (int)(s1 - s0);
        const long long d0 = t == 0 ? Lptr[p] : Uptr[nNodes - 1 - p];
        int* dcol = t == 0 ? Lcol : Ucol;
        VT* dval = t == 0 ? Lval : Uval;
        for (int a = 0; a < nE; a++) {
            const int pass = a >> 3, g = a & 7;
            const int nb = min(8, nE - 8 * pass);
            dval[(d0 + 8 * pass) * BILU_NB2 + (((r >> 1) * nb + g) * 8 + k) * 2 + (r & 1)] = (VT)bval[(s0 + a) * BILU_NB2 + lane];
            if (lane == 0) dcol[d0 + a] = bcol[s0 + a];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side object
// ---------------------------------------------------------------------------------------------------
struct NodeILU {
    bool ready = false;
    long long n = 0;
    int nNodes = 0, nPrimary = 0, nLevels = 0, nshift = 0, maxRow = 0;
    std::vector<unsigned char> h_late;  // per position: late node (extra unknowns of a cell, coupled to primary nodes only)
    long long nnzB = 0, nL = 0, nU = 0;
    bool fp32 = false;
    double windowLevels = 3.0;  // levels kept in flight by the sweeps (launch shape; measured optimum at 200 k cells)
    int launchGrid = 0, launchSleep = 0, launchPerXcd = 0;  // launch shape of the sweeps, fixed by bilu_setup (bilu_launch_shape)
    // host copies (tests / introspection)
    std::vector<int> h_nodeUnk, h_bcol, h_lvlPtr, h_natural;  // h_natural[p] = natural (cell-order) index of the node at position p
    std::vector<long long> h_bptr;
    DevBuf<int> nodeUnk, nodeOut, Lcol, Ucol;  // nodeOut: allocated only by bilu_setup_multi (else the view points at nodeUnk)
    std::vector<int> h_nodeOut;
    DevBuf<long long> Lptr, Uptr;
    DevBuf<double> Lval, Uval, invD, y, z;
    DevBuf<float> Lvalf, Uvalf;
    DevBuf<unsigned> ctrl;
    DevBuf<double> ym, zm;  // work vectors of the multi right-hand-side sweeps (allocated on first use)
    int mS = 0;
    BiluView view;
    double t_struct = 0, t_scatter = 0, t_factor = 0, t_pack = 0;
    long long factor_bytes() const { return (nL + nU) * BILU_NB2 * (fp32 ? 4 : 8) + (long long)nNodes * BILU_NB2 * 8; }
};

// reverse Cuthill-McKee ordering of the owned cells on the face-neighbour graph (jacMatReOrdering "rcm": the reference
// hands MATORDERINGRCM to the sub-domain ILU, DALinearEqn.C:238-290); start cells are pseudo-peripheral (two BFS sweeps)
inline std::vector<int> bilu_rcm_cells(const Mesh& m, const std::vector<char>& cellOwned, const double* startDir = nullptr) {
    const int nC = m.nC;
    std::vector<int> order, deg(nC, 0), queue;
    std::vector<char> seen(nC, 0);
    order.reserve(nC);
    for (int c = 0; c < nC; c++) for (int q = m.cc_ptr[c]; q < m.cc_ptr[c + 1]; q++) deg[c] += cellOwned[m.cc[q]] ? 1 : 0;
    auto bfs = [&](int start, std::vector<int>& out, std::vector<char>& mark) {
        const size_t first = out.size();
        out.push_back(start);
        mark[start] = 1;
        std::vector<int> nb;
        for (size_t head = first; head < out.size(); head++) {
            const int c = out[head];
            nb.clear();
            for (int q = m.cc_ptr[c]; q < m.cc_ptr[c + 1]; q++) { const int d = m.cc[q]; if (cellOwned[d] && !mark[d]) { mark[d] = 1; nb.push_back(d); } }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return deg[a] < deg[b] || (deg[a] == deg[b] && a < b); });
            out.insert(out.end(), nb.begin(), nb.end());
        }
    };
    std::vector<char> tmpMark(nC, 0);
    std::vector<int> tmp;
    for (int c0 = 0; c0 < nC; c0++) {
        if (!cellOwned[c0] || seen[c0]) continue;
        // pseudo-peripheral start: the last cell of a BFS from c0, then the last cell of a BFS from there; with a direction: the cell of
        // the component that lies furthest AGAINST it (the most upstream one)
        int start = c0;
        if (startDir) {
            tmp.clear();
            bfs(c0, tmp, tmpMark);
            double best = 1e300;
            for (int c : tmp) {
                tmpMark[c] = 0;
                const double key = m.cg[c].C[0] * startDir[0] + m.cg[c].C[1] * startDir[1] + m.cg[c].C[2] * startDir[2];
                if (key < best) { best = key; start = c; }
            }
        } else {
            for (int sweep = 0; sweep < 2; sweep++) {
                tmp.clear();
                bfs(start, tmp, tmpMark);
                start = tmp.back();
                for (int c : tmp) tmpMark[c] = 0;
            }
        }
        bfs(start, order, seen);
    }
    std::reverse(order.begin(), order.end());
    return order;
}

// Structure: nodes, node pattern, level order.  `states` = DAIndex state blocks, `owned` (per state, may be empty) /
// `cellOwned` restrict the preconditioner to this rank's unknowns, `reach` = stencil reach in cell rings.
inline void bilu_build_structure(const Mesh& m, const std::vector<StateDef>& states, long long n, const std::vector<unsigned char>& owned,
                                 const std::vector<char>& cellOwned, int reach, NodeILU& P, std::vector<int>& unkNode,
                                 std::vector<unsigned char>& unkSlot, std::vector<long long>& bptr, std::vector<long long>& bdiag,
                                 std::vector<int>& bcol, int nthr, int order = 0, const double* dir = nullptr) {
    const int nC = m.nC;
    // ---- unknowns cell by cell, packed into nodes of 8 slots
    std::vector<std::vector<int>> ownedFaces;
    bool hasFace = false;
    for (const StateDef& sd : states) if (sd.kind == KIND_FACE) hasFace = true;
    std::vector<int> of_ptr(nC + 1, 0), of;
    if (hasFace) {
        for (int f = 0; f < m.nF; f++) of_ptr[m.owner[f] + 1]++;
        for (int c = 0; c < nC; c++) of_ptr[c + 1] += of_ptr[c];
        of.resize(m.nF);
        std::vector<int> fill(of_ptr.begin(), of_ptr.end() - 1);
        for (int f = 0; f < m.nF; f++) of[fill[m.owner[f]]++] = f;
    }
    std::vector<int> nodeUnk0;           // natural node order
    std::vector<int> nodeCell_ptr{0}, nodeCell;  // node -> cells
    std::vector<int> cellNode_ptr(nC + 1, 0), cellNode;
    nodeUnk0.reserve((size_t)nC * 9);
    std::vector<long long> tmp;
    int fillSlots = BILU_NB;  // slots used in the current node (BILU_NB = closed)
    auto is_owned = [&](long long g) { return owned.empty() || owned[g]; };
    // cell order of the elimination: the mesh's own numbering, or RCM of the cell graph
    std::vector<int> cellOrder;
    // order: 0 the mesh's numbering, 1 reverse Cuthill-McKee, 2 Cuthill-McKee (1 backwards), 3 the mesh's numbering backwards, 4 / 5 below
    if (order == 1 || order == 2) cellOrder = bilu_rcm_cells(m, cellOwned);
    else { cellOrder.reserve(nC); for (int c = 0; c < nC; c++) if (cellOwned[c]) cellOrder.push_back(c); }
    if (order == 2 || order == 3) std::reverse(cellOrder.begin(), cellOrder.end());
    if (order == 6 || order == 7) {  // Cuthill-McKee levels grown from the most upstream cell: 6 downstream-wards, 7 upstream-wards
        const double xdir[3] = {1.0, 0.0, 0.0};
        cellOrder = bilu_rcm_cells(m, cellOwned, dir ? dir : xdir);
        if (order == 6) std::reverse(cellOrder.begin(), cellOrder.end());
    }
    if (order == 4 || order == 5) {  // 4 / 5: along / against the direction `dir` (the mean flow; default x), cell centres projected on it
        const double d0 = dir ? dir[0] : 1.0, d1 = dir ? dir[1] : 0.0, d2 = dir ? dir[2] : 0.0, sgn = order == 4 ? 1.0 : -1.0;
        std::vector<double> key(nC);
        for (int c = 0; c < nC; c++) key[c] = sgn * (m.cg[c].C[0] * d0 + m.cg[c].C[1] * d1 + m.cg[c].C[2] * d2);
        std::sort(cellOrder.begin(), cellOrder.end(), [&](int a, int b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
    }
    std::vector<std::pair<int, int>> cellNodePairs;  // (cell, node), cells in visiting order
    std::vector<int> lateUnk, lateCell;
    for (int c : cellOrder) {
        tmp.clear();
        for (const StateDef& sd : states) {
            if (sd.kind == KIND_VEC) { for (int k = 0; k < 3; k++) if (is_owned(sd.offset + 3LL * c + k)) tmp.push_back(sd.offset + 3LL * c + k); }
            else if (sd.kind == KIND_SCL) { if (is_owned(sd.offset + c)) tmp.push_back(sd.offset + c); }
        }
        for (const StateDef& sd : states)
            if (sd.kind == KIND_FACE) for (int q = of_ptr[c]; q < of_ptr[c + 1]; q++) if (is_owned(sd.offset + of[q])) tmp.push_back(sd.offset + of[q]);
        const size_t mc = tmp.size();
        if (mc == 0) continue;
        // PRIMARY node: the first <= 8 unknowns of the cell (its cell states and first owned faces); a cell that does not
        // fit into the open node starts a new one.  What is left (the extra owned faces of boundary cells) goes to LATE
        // nodes, numbered behind all primary nodes: they depend on primary nodes only, so the boundary cells do not
        // lengthen the dependent chain of the sweeps (see the pattern below).
        const size_t mp = std::min<size_t>(mc, BILU_NB);
        if ((size_t)(BILU_NB - fillSlots) < mp) fillSlots = BILU_NB;
        if (fillSlots == BILU_NB) {
            nodeUnk0.insert(nodeUnk0.end(), BILU_NB, -1);
            nodeCell_ptr.push_back(nodeCell_ptr.back());
            fillSlots = 0;
        }
        const int node = (int)nodeCell_ptr.size() - 2;
        nodeCell.push_back(c);
        nodeCell_ptr[node + 1]++;
        cellNodePairs.push_back({c, node});
        for (size_t k = 0; k < mp; k++) nodeUnk0[(size_t)node * BILU_NB + fillSlots++] = (int)tmp[k];
        for (size_t k = mp; k < mc; k++) { lateUnk.push_back((int)tmp[k]); lateCell.push_back(c); }
    }
    const int nPrimary = (int)nodeCell_ptr.size() - 1;
    for (size_t k = 0; k < lateUnk.size();) {
        const int c = lateCell[k];
        nodeUnk0.insert(nodeUnk0.end(), BILU_NB, -1);
        nodeCell_ptr.push_back(nodeCell_ptr.back());
        const int node = (int)nodeCell_ptr.size() - 2;
        nodeCell.push_back(c);
        nodeCell_ptr[node + 1]++;
        cellNodePairs.push_back({c, node});
        for (int slot = 0; slot < BILU_NB && k < lateUnk.size() && lateCell[k] == c; slot++, k++) nodeUnk0[(size_t)node * BILU_NB + slot] = lateUnk[k];
    }
    const int nN = (int)nodeCell_ptr.size() - 1;
    DAS_CHECK(nN > 0, DAS_ERR_INTERNAL, "preconditioner: no owned unknowns");
    for (const auto& pr : cellNodePairs) cellNode_ptr[pr.first + 1]++;
    for (int c = 0; c < nC; c++) cellNode_ptr[c + 1] += cellNode_ptr[c];
    cellNode.resize(cellNodePairs.size());
    {
        std::vector<int> fill(cellNode_ptr.begin(), cellNode_ptr.end() - 1);
        for (const auto& pr : cellNodePairs) cellNode[fill[pr.first]++] = pr.second;
    }
    // ---- node pattern: nodes of all cells within `reach` rings of the node's cells (symmetric by construction)
    std::vector<long long> rptr(nN + 1, 0);
    std::vector<std::vector<int>> rows(nN);
#pragma omp parallel num_threads(nthr)
    {
        std::vector<int> cmark(nC, -1), nmark(nN, -1), cur, nxt;
#pragma omp for schedule(dynamic, 256)
        for (int I = 0; I < nN; I++) {
            std::vector<int>& row = rows[I];
            cur.clear();
            for (int q = nodeCell_ptr[I]; q < nodeCell_ptr[I + 1]; q++) { const int c = nodeCell[q]; cmark[c] = I; cur.push_back(c); }
            // a late node is coupled to primary nodes (and itself) only: late-late couplings are dropped from the incomplete
            // factorisation, which keeps all late nodes mutually independent
            const bool lateI = I >= nPrimary;
            auto add_cell = [&](int c) {
                for (int q = cellNode_ptr[c]; q < cellNode_ptr[c + 1]; q++) {
                    const int J = cellNode[q];
                    if (lateI && J >= nPrimary && J != I) continue;
                    if (nmark[J] != I) { nmark[J] = I; row.push_back(J); }
                }
            };
            for (int c : cur) add_cell(c);
            for (int ring = 0; ring < reach; ring++) {
                nxt.clear();
                for (int c : cur)
                    for (int q = m.cc_ptr[c]; q < m.cc_ptr[c + 1]; q++) {
                        const int d = m.cc[q];
                        if (cmark[d] != I) { cmark[d] = I; nxt.push_back(d); if (cellOwned[d]) add_cell(d); }
                    }
                cur.swap(nxt);
            }
            std::sort(row.begin(), row.end());
        }
    }
    // ---- level sets of the lower-triangular dependency graph, level order = processing order
    std::vector<int> level(nN, 0);
    int nLv = 0;
    for (int I = 0; I < nN; I++) {
        int l = 0;
        for (int J : rows[I]) { if (J >= I) break; l = std::max(l, level[J] + 1); }
        level[I] = l;
        nLv = std::max(nLv, l + 1);
    }
    std::vector<int> lvlPtr(nLv + 1, 0), pos(nN);
    for (int I = 0; I < nN; I++) lvlPtr[level[I] + 1]++;
    for (int l = 0; l < nLv; l++) lvlPtr[l + 1] += lvlPtr[l];
    {
        std::vector<int> fill(lvlPtr.begin(), lvlPtr.end() - 1);
        for (int I = 0; I < nN; I++) pos[I] = fill[level[I]]++;
    }
    std::vector<int> inv(nN);
    for (int I = 0; I < nN; I++) inv[pos[I]] = I;
    // ---- permuted block CSR
    bptr.assign(nN + 1, 0);
    for (int p = 0; p < nN; p++) bptr[p + 1] = bptr[p] + (long long)rows[inv[p]].size();
    bcol.resize((size_t)bptr[nN]);
    bdiag.assign(nN, -1);
    int maxRow = 0;
#pragma omp parallel for schedule(static) reduction(max : maxRow) num_threads(nthr)
    for (int p = 0; p < nN; p++) {
        const std::vector<int>& row = rows[inv[p]];
        int* dst = bcol.data() + bptr[p];
        for (size_t k = 0; k < row.size(); k++) dst[k] = pos[row[k]];
        std::sort(dst, dst + row.size());
        bdiag[p] = bptr[p] + (std::lower_bound(dst, dst + row.size(), p) - dst);
        maxRow = std::max(maxRow, (int)row.size());
    }
    P.h_nodeUnk.assign((size_t)nN * BILU_NB, -1);
    unkNode.assign(n, -1);
    unkSlot.assign(n, 0);
    for (int p = 0; p < nN; p++)
        for (int k = 0; k < BILU_NB; k++) {
            const int gidx = nodeUnk0[(size_t)inv[p] * BILU_NB + k];
            P.h_nodeUnk[(size_t)p * BILU_NB + k] = gidx;
            if (gidx >= 0) { unkNode[gidx] = p; unkSlot[gidx] = (unsigned char)k; }
        }
    P.n = n; P.nNodes = nN; P.nLevels = nLv; P.nnzB = bptr[nN]; P.maxRow = maxRow; P.nPrimary = nPrimary;
    P.h_late.assign(nN, 0);
    for (int p = 0; p < nN; p++) P.h_late[p] = inv[p] >= nPrimary ? 1 : 0;
    P.h_bptr = bptr; P.h_bcol = bcol; P.h_lvlPtr = lvlPtr; P.h_natural = inv;
}

struct NodeILU;
inline void bilu_launch_shape(NodeILU& P, hipStream_t st);

// host structure of a factorisation handed to the numeric part: permuted block CSR + one (unknown -> node, slot) map per block of unknowns
struct BiluStructRef {
    const std::vector<long long>* bptr;
    const std::vector<long long>* bdiag;
    const std::vector<int>* bcol;
    std::vector<const std::vector<int>*> unkNode;
    std::vector<const std::vector<unsigned char>*> unkSlot;
};
inline void bilu_numeric(long long n, const BiluStructRef& S, bool fp32, long long An, const long long* d_rp, const int* d_ci, const double* d_av, hipStream_t st, NodeILU& P,
                         bool debug, bool transpose, double diagScale, long long shiftExLo, long long shiftExHi, long long shiftEnd, double t0);

// Numeric setup on the device from the assembled PC matrix (device CSR, rows = states).
inline void bilu_setup(const Mesh& m, const std::vector<StateDef>& states, long long n, const std::vector<unsigned char>& owned, int reach,
                       bool fp32, long long An, const long long* d_rp, const int* d_ci, const double* d_av, hipStream_t st, NodeILU& P,
                       bool debug, int nthr, int order, bool transpose = false, double diagScale = 1.0, long long shiftExLo = 0,
                       long long shiftExHi = 0, long long shiftEnd = (long long)1 << 62, const double* dir = nullptr) {
    const double t0 = wall_seconds();
    std::vector<char> cellOwned(m.nC, 1);
    if (!owned.empty()) {
        const StateDef& s0 = states[0];
        const int stride = s0.kind == KIND_VEC ? 3 : 1;
        for (int c = 0; c < m.nC; c++) cellOwned[c] = owned[s0.offset + (long long)stride * c] ? 1 : 0;
    }
    std::vector<int> unkNode, bcol;
    std::vector<unsigned char> unkSlot;
    std::vector<long long> bptr, bdiag;
    bilu_build_structure(m, states, n, owned, cellOwned, reach, P, unkNode, unkSlot, bptr, bdiag, bcol, std::max(1, nthr), order, dir);
    BiluStructRef S{&bptr, &bdiag, &bcol, {&unkNode}, {&unkSlot}};
    bilu_numeric(n, S, fp32, An, d_rp, d_ci, d_av, st, P, debug, transpose, diagScale, shiftExLo, shiftExHi, shiftEnd, t0);
}

inline void bilu_numeric(long long n, const BiluStructRef& S, bool fp32, long long An, const long long* d_rp, const int* d_ci, const double* d_av, hipStream_t st, NodeILU& P,
                         bool debug, bool transpose, double diagScale, long long shiftExLo, long long shiftExHi, long long shiftEnd, double t0) {
    const int nN = P.nNodes;
    P.fp32 = fp32;
    P.nodeUnk.upload(P.h_nodeUnk);
    DevBuf<int> d_unkNode, d_bcol;
    DevBuf<unsigned char> d_unkSlot;
    DevBuf<long long> d_bptr, d_bdiag;
    const std::vector<int>& bcol = *S.bcol;
    const std::vector<long long>&bptr = *S.bptr, &bdiag = *S.bdiag;
    DevBuf<unsigned char> d_late;
    d_late.upload(P.h_late);
    d_bptr.upload(bptr); d_bdiag.upload(bdiag); d_bcol.upload(bcol);
    DevBuf<double> bval((size_t)P.nnzB * BILU_NB2);
    DevBuf<unsigned long long> d_dropped(1);
    DevBuf<int> d_nshift(1);
    DAS_HIP(hipMemsetAsync(bval.p, 0, bval.n * sizeof(double), st));
    DAS_HIP(hipMemsetAsync(d_dropped.p, 0, sizeof(unsigned long long), st));
    DAS_HIP(hipMemsetAsync(d_nshift.p, 0, sizeof(int), st));
    P.t_struct = wall_seconds() - t0;
    double t1 = wall_seconds();
    for (size_t q = 0; q < S.unkNode.size(); q++) {  // one scatter per block of unknowns (one; K for a multi-block factorisation)
        d_unkNode.upload(*S.unkNode[q]); d_unkSlot.upload(*S.unkSlot[q]);
        hipLaunchKernelGGL(k_bilu_scatter, dim3((unsigned)((An + 15) / 16)), dim3(256), 0, st, An, d_rp, d_ci, d_av, d_unkNode.p, d_unkSlot.p, d_bptr.p,
                           d_bcol.p, d_late.p, bval.p, d_dropped.p, transpose ? 1 : 0, diagScale, shiftExLo, shiftExHi, shiftEnd);
        DAS_HIP(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_bilu_pad_diag, dim3((unsigned)(((long long)nN * BILU_NB + 255) / 256)), dim3(256), 0, st, nN, P.nodeUnk.p, d_bdiag.p, bval.p);
    DAS_HIP(hipGetLastError());
    unsigned long long dropped = 0;
    DAS_HIP(hipMemcpyAsync(&dropped, d_dropped.p, sizeof(dropped), hipMemcpyDeviceToHost, st));
    DAS_HIP(hipStreamSynchronize(st));
    DAS_CHECK(dropped == 0, DAS_ERR_INTERNAL, "preconditioner: " + std::to_string(dropped) + " matrix entries fall outside the node pattern (stencil reach too small)");
    P.t_scatter = wall_seconds() - t1;
    t1 = wall_seconds();
    // ---- numeric factorisation, one launch per level
    P.invD.alloc((size_t)nN * BILU_NB2);
    for (int l = 0; l < P.nLevels; l++) {
        const int a = P.h_lvlPtr[l], bnd = P.h_lvlPtr[l + 1];
        if (bnd > a)
            hipLaunchKernelGGL(k_bilu_factor, dim3((unsigned)((bnd - a + 3) / 4)), dim3(256), 0, st, a, bnd, d_bptr.p, d_bdiag.p, d_bcol.p, bval.p, P.invD.p,
                               d_nshift.p);
    }
    DAS_HIP(hipGetLastError());
    DAS_HIP(hipMemcpyAsync(&P.nshift, d_nshift.p, sizeof(int), hipMemcpyDeviceToHost, st));
    DAS_HIP(hipStreamSynchronize(st));
    P.t_factor = wall_seconds() - t1;
    t1 = wall_seconds();
    // ---- sweep streams
    std::vector<long long> Lptr(nN + 1, 0), Uptr(nN + 1, 0);
    for (int p = 0; p < nN; p++) Lptr[p + 1] = Lptr[p] + (bdiag[p] - bptr[p]);
    for (int q = 0; q < nN; q++) { const int p = nN - 1 - q; Uptr[q + 1] = Uptr[q] + (bptr[p + 1] - bdiag[p] - 1); }
    P.nL = Lptr[nN]; P.nU = Uptr[nN];
    P.Lptr.upload(Lptr); P.Uptr.upload(Uptr);
    P.Lcol.alloc(std::max<long long>(P.nL, 1)); P.Ucol.alloc(std::max<long long>(P.nU, 1));
    if (fp32) {
        P.Lvalf.alloc((size_t)std::max<long long>(P.nL, 1) * BILU_NB2); P.Uvalf.alloc((size_t)std::max<long long>(P.nU, 1) * BILU_NB2);
        P.Lval.release(); P.Uval.release();
        hipLaunchKernelGGL(k_bilu_pack<float>, dim3((unsigned)((nN + 3) / 4)), dim3(256), 0, st, nN, d_bptr.p, d_bdiag.p, d_bcol.p, bval.p, P.Lptr.p,
                           P.Uptr.p, P.Lcol.p, P.Ucol.p, P.Lvalf.p, P.Uvalf.p);
    } else {
        P.Lval.alloc((size_t)std::max<long long>(P.nL, 1) * BILU_NB2); P.Uval.alloc((size_t)std::max<long long>(P.nU, 1) * BILU_NB2);
        P.Lvalf.release(); P.Uvalf.release();
        hipLaunchKernelGGL(k_bilu_pack<double>, dim3((unsigned)((nN + 3) / 4)), dim3(256), 0, st, nN, d_bptr.p, d_bdiag.p, d_bcol.p, bval.p, P.Lptr.p,
                           P.Uptr.p, P.Lcol.p, P.Ucol.p, P.Lval.p, P.Uval.p);
    }
    DAS_HIP(hipGetLastError());
    P.y.alloc((size_t)nN * BILU_NB); P.z.alloc((size_t)nN * BILU_NB);
    P.ctrl.alloc(BILU_CTRL_SIZE);
    DAS_HIP(hipMemsetAsync(P.ctrl.p, 0, BILU_CTRL_SIZE * sizeof(unsigned), st));
    DAS_HIP(hipStreamSynchronize(st));
    P.t_pack = wall_seconds() - t1;
    P.view.nNodes = nN; P.view.nodeUnk = P.nodeUnk.p;
    if (!P.h_nodeOut.empty()) P.nodeOut.upload(P.h_nodeOut);
    P.view.nodeOut = P.h_nodeOut.empty() ? P.nodeUnk.p : P.nodeOut.p;
    P.view.ptr[0] = P.Lptr.p; P.view.ptr[1] = P.Uptr.p; P.view.col[0] = P.Lcol.p; P.view.col[1] = P.Ucol.p;
    P.view.val[0] = P.Lval.p; P.view.val[1] = P.Uval.p; P.view.valf[0] = P.Lvalf.p; P.view.valf[1] = P.Uvalf.p;
    P.view.invD = P.invD.p; P.view.y = P.y.p; P.view.z = P.z.p; P.view.ctrl = P.ctrl.p;
    bilu_launch_shape(P, st);
    P.ready = true;
    if (debug)
        fprintf(stderr, "[dafoam_amd] node-block ILU(0): %d nodes (%.3f slots/unknown), %lld blocks (%.1f per node, max %d), %d levels, %d shifted pivots, "
                        "factor %.2f GB%s; structure %.2f s, scatter %.3f s, factorise %.3f s, pack %.3f s\n", nN, (double)nN * BILU_NB / (double)n, P.nnzB,
                (double)P.nnzB / nN, P.maxRow, P.nLevels, P.nshift, P.factor_bytes() / 1e9, fp32 ? " (fp32)" : "", P.t_struct, P.t_scatter, P.t_factor,
                P.t_pack);
}

// K sub-domain factorisations (restricted additive Schwarz inside one GPU) as ONE structure: the node graphs of the blocks are disjoint, so
// their level sets interleave - level l of the merged structure holds level l of every block - and ONE pair of sweeps runs all blocks at
// once: max(levels) dependent hops instead of their sum, sum(nodes) / max(levels) nodes per level to fill the device.  An unknown of an
// overlap ring sits in a node of every block that reaches it; all copies read the right-hand side, only the owner block's copy writes the
// solution (nodeOut).  masks[b]: unknowns of block b (owned + overlap), orders[b]: its elimination order, subOf[unknown]: the owner block.
inline void bilu_setup_multi(const Mesh& m, const std::vector<StateDef>& states, long long n, const std::vector<const std::vector<unsigned char>*>& masks,
                             const std::vector<int>& orders, const std::vector<unsigned char>& subOf, int reach, bool fp32, long long An, const long long* d_rp,
                             const int* d_ci, const double* d_av, hipStream_t st, NodeILU& P, bool debug, int nthr, const double* dir) {
    const double t0 = wall_seconds();
    const int K = (int)masks.size();
    struct Blk { NodeILU S; std::vector<int> unkNode, bcol, newPos; std::vector<unsigned char> unkSlot; std::vector<long long> bptr, bdiag; };
    std::vector<Blk> B(K);
    int nLv = 0;
    long long nN = 0, nnzB = 0;
    for (int b = 0; b < K; b++) {
        const std::vector<unsigned char>& mask = *masks[b];
        std::vector<char> cellOwned(m.nC, 1);
        const StateDef& s0 = states[0];
        const int stride = s0.kind == KIND_VEC ? 3 : 1;
        for (int c = 0; c < m.nC; c++) cellOwned[c] = mask[s0.offset + (long long)stride * c] ? 1 : 0;
        bilu_build_structure(m, states, n, mask, cellOwned, reach, B[b].S, B[b].unkNode, B[b].unkSlot, B[b].bptr, B[b].bdiag, B[b].bcol, std::max(1, nthr), orders[b], dir);
        nLv = std::max(nLv, B[b].S.nLevels);
        nN += B[b].S.nNodes;
        nnzB += B[b].S.nnzB;
    }
    DAS_CHECK(nN < (1LL << 31) && nnzB < (1LL << 40), DAS_ERR_INTERNAL, "multi-block preconditioner too large");
    // ---- merged processing order: level-major, block after block inside a level
    std::vector<int> lvlPtr(nLv + 1, 0);
    for (int b = 0; b < K; b++) for (int l = 0; l < B[b].S.nLevels; l++) lvlPtr[l + 1] += B[b].S.h_lvlPtr[l + 1] - B[b].S.h_lvlPtr[l];
    for (int l = 0; l < nLv; l++) lvlPtr[l + 1] += lvlPtr[l];
    {
        std::vector<int> cursor(lvlPtr.begin(), lvlPtr.end() - 1);
        for (int b = 0; b < K; b++) {
            B[b].newPos.resize(B[b].S.nNodes);
            for (int l = 0; l < B[b].S.nLevels; l++) for (int p = B[b].S.h_lvlPtr[l]; p < B[b].S.h_lvlPtr[l + 1]; p++) B[b].newPos[p] = cursor[l]++;
        }
    }
    std::vector<long long> bptr(nN + 1, 0), bdiag(nN, -1);
    for (int b = 0; b < K; b++) for (int p = 0; p < B[b].S.nNodes; p++) bptr[B[b].newPos[p] + 1] = B[b].bptr[p + 1] - B[b].bptr[p];
    for (long long p = 0; p < nN; p++) bptr[p + 1] += bptr[p];
    std::vector<int> bcol((size_t)bptr[nN]);
    P = NodeILU();
    P.h_nodeUnk.assign((size_t)nN * BILU_NB, -1);
    P.h_nodeOut.assign((size_t)nN * BILU_NB, -1);
    P.h_late.assign(nN, 0);
    P.h_natural.assign(nN, 0);
    int maxRow = 0, nPrimary = 0;
    long long natOff = 0;
    for (int b = 0; b < K; b++) {
        Blk& Q = B[b];
#pragma omp parallel for schedule(static) num_threads(std::max(1, nthr))
        for (int p = 0; p < Q.S.nNodes; p++) {
            const long long np = Q.newPos[p];
            int* dst = bcol.data() + bptr[np];
            const long long r0 = Q.bptr[p], len = Q.bptr[p + 1] - r0;
            for (long long q = 0; q < len; q++) dst[q] = Q.newPos[Q.bcol[r0 + q]];  // stays ascending: positions of one block keep their order
            bdiag[np] = bptr[np] + (Q.bdiag[p] - r0);
            for (int k = 0; k < BILU_NB; k++) {
                const int gi = Q.S.h_nodeUnk[(size_t)p * BILU_NB + k];
                P.h_nodeUnk[(size_t)np * BILU_NB + k] = gi;
                P.h_nodeOut[(size_t)np * BILU_NB + k] = (gi >= 0 && subOf[gi] == (unsigned char)b) ? gi : -1;
            }
            P.h_late[np] = Q.S.h_late[p];
            P.h_natural[np] = (int)(natOff + Q.S.h_natural[p]);
        }
        for (long long g = 0; g < n; g++) if (Q.unkNode[g] >= 0) Q.unkNode[g] = Q.newPos[Q.unkNode[g]];
        maxRow = std::max(maxRow, Q.S.maxRow);
        nPrimary += Q.S.nPrimary;
        natOff += Q.S.nNodes;
        Q.S = NodeILU();
        std::vector<int>().swap(Q.bcol);
        std::vector<long long>().swap(Q.bptr);
        std::vector<long long>().swap(Q.bdiag);
    }
    P.n = n; P.nNodes = (int)nN; P.nLevels = nLv; P.nnzB = bptr[nN]; P.maxRow = maxRow; P.nPrimary = nPrimary;
    P.h_bptr = bptr; P.h_bcol = bcol; P.h_lvlPtr = lvlPtr;
    BiluStructRef S{&bptr, &bdiag, &bcol, {}, {}};
    for (int b = 0; b < K; b++) { S.unkNode.push_back(&B[b].unkNode); S.unkSlot.push_back(&B[b].unkSlot); }
    bilu_numeric(n, S, fp32, An, d_rp, d_ci, d_av, st, P, debug, false, 1.0, 0, 0, (long long)1 << 62, t0);
}

// Which XCC ids do the workgroups of a launch on this device see?  One tiny launch per process (cached): 256 workgroups OR
// 1 << XCC_ID into a mask.  Returns the number of XCDs if the ids are exactly 0..7 (an unpartitioned MI300X / MI355X), else 0
// (partition modes, other parts): the per-XCD ticket scheme of k_bilu_sweep is then never used.
__global__ void k_bilu_xcd_probe(unsigned* mask) {
    if (threadIdx.x == 0) atomicOr(mask, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));
}
inline int bilu_xcd_probe(hipStream_t st) {
    static int cached = -1;
    if (cached >= 0) return cached;
    DevBuf<unsigned> mask(1);
    DAS_HIP(hipMemsetAsync(mask.p, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_bilu_xcd_probe, dim3(256), dim3(64), 0, st, mask.p);
    unsigned h = 0u;
    DAS_HIP(hipMemcpyAsync(&h, mask.p, sizeof(h), hipMemcpyDeviceToHost, st));
    DAS_HIP(hipStreamSynchronize(st));
    cached = (h == 0xFFu) ? 8 : 0;
    return cached;
}

// launch shape of the sweeps: as many workgroups as keep a few levels in flight (a wave far ahead of the front only
// spins and loads the memory system).  DAS_BILU_WGS / DAS_BILU_SLEEP / DAS_BILU_XCD override (tuning runs).  Computed once
// per factorisation (bilu_setup) and cached in the NodeILU.
inline void bilu_launch_shape(NodeILU& P, hipStream_t st) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int waves = BILU_WG / 64;
    const double perLevel = (double)P.nNodes / std::max(1, P.nLevels);
    int grid = (int)std::min<double>(cus * 2.0 * (BILU_OCC * 4 / waves), std::max(8.0, P.windowLevels * perLevel / waves));  // <= 2 x resident
    int sleepReps = 0;
    if (const char* e = getenv("DAS_BILU_WGS")) grid = std::max(1, atoi(e));
    if (const char* e = getenv("DAS_BILU_SLEEP")) sleepReps = std::max(0, atoi(e));
    const long long tickets = ((long long)P.nNodes + waves - 1) / waves;
    grid = (int)std::min<long long>(grid, tickets + 1);
    // per-XCD ticket counters only where every XCD is certain to run workgroups of the launch (see k_bilu_sweep)
    int perXcd = (BILU_XCD_TICKETS && grid >= 64 && bilu_xcd_probe(st) == 8) ? 1 : 0;
    if (const char* e = getenv("DAS_BILU_XCD")) perXcd = (atoi(e) != 0 && grid >= 64 && bilu_xcd_probe(st) == 8) ? 1 : 0;
    P.launchGrid = grid; P.launchSleep = sleepReps; P.launchPerXcd = perXcd;
}

// out = (LU)^-1 b on the owned unknowns (entries of `out` outside the preconditioner's unknowns are not touched)
inline void bilu_apply(NodeILU& P, const double* b, double* out, hipStream_t st) {
    const long long nslots = (long long)P.nNodes * BILU_NB;
    hipLaunchKernelGGL(k_bilu_reset, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, nslots, P.y.p, P.z.p, P.ctrl.p);
    const int grid = P.launchGrid, sl = P.launchSleep, px = P.launchPerXcd;
    if (P.fp32) {
        hipLaunchKernelGGL((k_bilu_sweep<float, false>), dim3(grid), dim3(BILU_WG), 0, st, P.view, b, out, sl, px);
        hipLaunchKernelGGL((k_bilu_sweep<float, true>), dim3(grid), dim3(BILU_WG), 0, st, P.view, b, out, sl, px);
    } else {
        hipLaunchKernelGGL((k_bilu_sweep<double, false>), dim3(grid), dim3(BILU_WG), 0, st, P.view, b, out, sl, px);
        hipLaunchKernelGGL((k_bilu_sweep<double, true>), dim3(grid), dim3(BILU_WG), 0, st, P.view, b, out, sl, px);
    }
}

// S right-hand sides through one pair of sweeps (S = 2 or 4): b, out column-major with leading dimension ld
template <int S>
inline void bilu_apply_multi_s(NodeILU& P, const double* b, double* out, long long ld, hipStream_t st) {
    const long long nslots = (long long)P.nNodes * BILU_NB * S;
    if (P.mS < S || P.ym.n < (size_t)nslots) { P.ym.alloc((size_t)P.nNodes * BILU_NB * 4); P.zm.alloc((size_t)P.nNodes * BILU_NB * 4); P.mS = 4; }
    hipLaunchKernelGGL(k_bilu_reset, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, nslots, P.ym.p, P.zm.p, P.ctrl.p);
    const int grid = P.launchGrid, sl = P.launchSleep, px = P.launchPerXcd;
    if (P.fp32) {
        hipLaunchKernelGGL((k_bilu_sweep_m<float, false, S>), dim3(grid), dim3(BILU_WG), 0, st, P.view, P.ym.p, P.zm.p, b, out, ld, sl, px);
        hipLaunchKernelGGL((k_bilu_sweep_m<float, true, S>), dim3(grid), dim3(BILU_WG), 0, st, P.view, P.ym.p, P.zm.p, b, out, ld, sl, px);
    } else {
        hipLaunchKernelGGL((k_bilu_sweep_m<double, false, S>), dim3(grid), dim3(BILU_WG), 0, st, P.view, P.ym.p, P.zm.p, b, out, ld, sl, px);
        hipLaunchKernelGGL((k_bilu_sweep_m<double, true, S>), dim3(grid), dim3(BILU_WG), 0, st, P.view, P.ym.p, P.zm.p, b, out, ld, sl, px);
    }
}
// any number of right-hand sides: groups of 4, then 2, then 1
inline void bilu_apply_multi(NodeILU& P, const double* b, double* out, long long ld, int nrhs, hipStream_t st) {
    int r = 0;
    for (; r + 4 <= nrhs; r += 4) bilu_apply_multi_s<4>(P, b + (long long)r * ld, out + (long long)r * ld, ld, st);
    for (; r + 2 <= nrhs; r += 2) bilu_apply_multi_s<2>(P, b + (long long)r * ld, out + (long long)r * ld, ld, st);
    for (; r < nrhs; r++) bilu_apply(P, b + (long long)r * ld, out + (long long)r * ld, st);
}

// abort flag of the sweeps (set when a bounded spin ran out): checked by the solver at its synchronisation points
inline bool bilu_aborted(NodeILU& P, hipStream_t st) {
    unsigned c = 0u;
    DAS_HIP(hipMemcpyAsync(&c, P.ctrl.p + BILU_CTRL_ABORT, sizeof(c), hipMemcpyDeviceToHost, st));
    DAS_HIP(hipStreamSynchronize(st));
    return c != 0u;
}

// a new solve starts with a clear abort flag (a sweep that timed out once must not disable the preconditioner for good)
inline void bilu_clear_abort(NodeILU& P, hipStream_t st) { DAS_HIP(hipMemsetAsync(P.ctrl.p + BILU_CTRL_ABORT, 0, sizeof(unsigned), st)); }

}  // namespace das
