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
#include <chrono>
#include <cstdlib>
#include <thread>

#include "das_common.hpp"

namespace das {

constexpr int COLOR_MAXW = 64;  // up to 64 x 64 = 4096 colours (bitmap words per net: 8, doubled on overflow)

struct ColorView {
    long long nGroups;
    const long long* gstart;   // nGroups+1: first column of every group (columns of a group share their kept-row list)
    const long long* cptr;     // n+1: kept rows ("nets") of a column
    const int* crow;           // net ids (positions in the kept-row list)
    const int* cpos;           // position of the column inside that net's (ascending) column list
    int W;                     // bitmap words per net
    unsigned long long* F;     // nNets x W: colours used by the columns of a net that are coloured so far
    unsigned* done;            // nNets: how many columns of a net are coloured so far
    int* colors;               // n, -1 = not coloured yet
    unsigned* ctrl;            // [0] ticket (4 groups each), [1] abort (1) / overflow: more than 64 W colours (2)
    unsigned* host;            // pinned HOST memory seen by the device: [0] progress (tickets drawn, written now and then), [1] stop request of the watchdog
};

// The serial first-fit, column j = smallest colour not used by a lower-numbered column sharing a kept row, as a data-flow
// computation over NET BITMAPS.  Round 2 gathered the colours of the whole distance-2 neighbourhood with multiplicity (~63
// nets x ~275 columns = 17 k words per column, 2e11 gathers at 2 M cells: 9.8 s, work-bound).  Here a net carries the bitmap
// of the colours its columns use and a counter of how many of its columns are coloured.  Inside a net the columns are
// coloured in ascending order (each waits for the ones before it), so column j at position pos of net r may proceed when
// done[r] == pos, and the bitmap then holds exactly the colours of its lower-numbered columns: the forbidden set of j is the
// OR of ~63 bitmaps (63 x W words instead of 17 k), its publication 63 atomicOr + 63 counter increments.  One wavefront per
// column group, four groups per ticket, ticket order = column order (a waiting wave only waits for groups that are already
// running); bounded spins.  The result is bit-identical to the host's serial sweep (tested).
__global__ __launch_bounds__(256) void k_color_firstfit(ColorView P) {
    __shared__ unsigned sh_ticket;
    __shared__ unsigned long long fbs[4][COLOR_MAXW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W = P.W;
    for (;;) {
        __syncthreads();  // the previous groups are done with fbs / sh_ticket
        if (threadIdx.x == 0) {  // (an aborted run - watchdog or a dead wave - hands out no further work)
            unsigned t = 0xFFFFFFFFu;
            if (__hip_atomic_load(&P.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                t = atomicAdd(&P.ctrl[0], 1u);
                if ((t & 1023u) == 0u) {
                    // every 1024th ticket talks to the host: progress out, stop request in (copies on a side stream do not run
                    // beside this persistent kernel - measured: the first one returned after 26 s)
                    __hip_atomic_store(&P.host[0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (__hip_atomic_load(&P.host[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
                        __hip_atomic_store(&P.ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        t = 0xFFFFFFFFu;
                    }
                }
            }
            sh_ticket = t;
        }
        if (lane < W) fbs[wave][lane] = 0ull;
        __syncthreads();
        const long long g = (long long)sh_ticket * 4 + wave;
        if ((long long)sh_ticket * 4 >= P.nGroups) return;
        if (g >= P.nGroups) continue;
        const long long j0 = P.gstart[g];
        const int m = (int)(P.gstart[g + 1] - j0);
        const long long q0 = P.cptr[j0], q1 = P.cptr[j0 + 1];
        // 1. wait until every net has coloured the columns in front of this group
        bool dead = false;
        for (long long q = q0 + lane; q < q1; q += 64) {
            const int r = P.crow[q];
            const unsigned need = (unsigned)P.cpos[q];
            unsigned spins = 0;
            while (__hip_atomic_load(&P.done[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023u) == 0u &&
                    (__hip_atomic_load(&P.ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 24))) {
                    __hip_atomic_store(&P.ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    break;
                }
            }
        }
        if (__any(dead)) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // 2. forbidden set = OR of the nets' bitmaps (read past the L1 / the other XCDs' L2: agent scope)
        for (long long q = q0 + lane; q < q1; q += 64) {
            const unsigned long long* f = P.F + (long long)P.crow[q] * W;
            for (int w = 0; w < W; w++) {
                const unsigned long long x = __hip_atomic_load(&f[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x) atomicOr(&fbs[wave][w], x);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // 3. the members of the group take the smallest free colours one after the other (lane w owns word w)
        unsigned long long mine = lane < W ? fbs[wave][lane] : ~0ull;
        int cs[8];
        const int mm = m < 8 ? m : 8;
        bool over = false;
        for (int i = 0; i < mm; i++) {
            const unsigned long long wz = ~mine;
            int best = wz ? (lane * 64 + __builtin_ctzll(wz)) : (1 << 30);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
            if (best >= 64 * W) { over = true; best = 64 * W - 1; }
            if (lane == (best >> 6)) mine |= 1ull << (best & 63);
            cs[i] = best;
        }
        if (over || m > 8) { if (lane == 0) __hip_atomic_store(&P.ctrl[1], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        for (int i = 0; i < mm; i++) if (lane == 0) P.colors[j0 + i] = cs[i];
        // 4. publish: the colours into every net's bitmap, THEN the counters (release order)
        for (long long q = q0 + lane; q < q1; q += 64) {
            unsigned long long* f = P.F + (long long)P.crow[q] * W;
            for (int i = 0; i < mm; i++)
                __hip_atomic_fetch_or(&f[cs[i] >> 6], 1ull << (cs[i] & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        for (long long q = q0 + lane; q < q1; q += 64)
            __hip_atomic_fetch_add(&P.done[P.crow[q]], (unsigned)mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the kernel driver shared by both front ends: device arrays in, colours out.  Returns 1 on success, 0 if the kernel could not
// be used (more than 4096 colours, a dead wave), -1 if the WATCHDOG stopped it: the data-flow first-fit is as parallel as the
// column numbering lets it be - a wavefront sweep on a lexicographic hex numbering (3 s at 2 M cells), but a numbering whose
// rows wrap around (an O-grid: the first cell of ring j+1 neighbours the LAST cells of ring j) serialises it completely
// (measured: 63 s at 2 M cells).  The host watches the ticket counter through a side stream; when the projected run time
// exceeds max(10 s, 1 us per column) it raises the abort flag and the caller switches to the order-independent algorithm.
inline int color_firstfit_run(long long n, long long nNets, const std::vector<long long>& gstartHost, const long long* d_cptr, const int* d_crow,
                              const int* d_cpos, std::vector<int>& colors, hipStream_t st) {
    const long long nGroups = (long long)gstartHost.size() - 1;
    DevBuf<long long> d_gstart;
    d_gstart.upload(gstartHost);
    DevBuf<int> d_colors(n);
    DevBuf<unsigned> d_ctrl(2), d_done(std::max<long long>(1, nNets));
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    long long wgs = (long long)cus * 8;
    if (const char* e = getenv("DAS_COLOR_WGS")) wgs = std::max(1, atoi(e));
    const int grid = (int)std::min<long long>(wgs, (nGroups + 3) / 4 + 1);
    double limit = std::max(10.0, 1.0e-6 * (double)n);
    if (const char* e = getenv("DAS_COLOR_LIMIT")) limit = atof(e);
    const double totalTickets = (double)((nGroups + 3) / 4);
    struct Side {  // an event to poll and two words of pinned host memory the kernel can see
        hipEvent_t ev = nullptr; unsigned* pin = nullptr;
        ~Side() { if (ev) (void)hipEventDestroy(ev); if (pin) (void)hipHostFree(pin); }
    } side;
    DAS_HIP(hipEventCreateWithFlags(&side.ev, hipEventDisableTiming));
    DAS_HIP(hipHostMalloc((void**)&side.pin, 4 * sizeof(unsigned), hipHostMallocMapped));
    unsigned* d_host = nullptr;
    DAS_HIP(hipHostGetDevicePointer((void**)&d_host, side.pin, 0));
    for (int W = 8; W <= COLOR_MAXW; W *= 2) {
        DevBuf<unsigned long long> d_F((size_t)std::max<long long>(1, nNets) * W);
        DAS_HIP(hipMemsetAsync(d_F.p, 0, d_F.n * sizeof(unsigned long long), st));
        DAS_HIP(hipMemsetAsync(d_done.p, 0, d_done.n * sizeof(unsigned), st));
        DAS_HIP(hipMemsetAsync(d_colors.p, 0xff, n * sizeof(int), st));
        DAS_HIP(hipMemsetAsync(d_ctrl.p, 0, 2 * sizeof(unsigned), st));
        ColorView V{nGroups, d_gstart.p, d_cptr, d_crow, d_cpos, W, d_F.p, d_done.p, d_colors.p, d_ctrl.p, d_host};
        volatile unsigned* pin = side.pin;
        pin[0] = 0u; pin[1] = 0u;
        DAS_HIP(hipStreamSynchronize(st));
        hipLaunchKernelGGL(k_color_firstfit, dim3(grid), dim3(256), 0, st, V);
        DAS_HIP(hipGetLastError());
        DAS_HIP(hipEventRecord(side.ev, st));
        const double t0 = wall_seconds();
        bool stopped = false;
        while (hipEventQuery(side.ev) == hipErrorNotReady) {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            const double el = wall_seconds() - t0;
            if (el < std::min(1.0, 0.5 * limit) || stopped) continue;
            const double projected = el * totalTickets / std::max(1.0, (double)pin[0]);
            if (projected > limit) {
                pin[1] = 1u;
                stopped = true;
                fprintf(stderr, "[dafoam_amd] data-flow first-fit colouring stopped after %.1f s (%.0f of %.0f tickets: %.0f s projected, limit %.0f s) - this column numbering "
                                "serialises it\n", el, (double)side.pin[0], totalTickets, projected, limit);
            }
        }
        unsigned ctrl[2] = {0, 0};
        DAS_HIP(hipMemcpyAsync(ctrl, d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost, st));
        DAS_HIP(hipStreamSynchronize(st));
        if (stopped) return -1;
        if (ctrl[1] == 2u) continue;  // more than 64 W colours: wider bitmaps
        if (ctrl[1] != 0u) return 0;
        colors.resize(n);
        d_colors.download(colors.data(), n);
        return 1;
    }
    return 0;
}
// groups from start flags: maximal runs of columns with identical net lists, cut at 8 members
inline std::vector<long long> color_groups_from_flags(long long n, const std::vector<unsigned char>& isStart) {
    std::vector<long long> gstart;
    gstart.reserve(n / 2 + 2);
    long long run = 0;
    for (long long j = 0; j < n; j++) {
        if (isStart[j] || run == 8) { gstart.push_back(j); run = 0; }
        run++;
    }
    gstart.push_back(n);
    return gstart;
}

// colours of the serial first-fit on the device.  keep/cptr/crow/cpos: the kept (non-dominated) rows and the CSC over them with
// the position of every column inside its nets (das_jaccon.cpp); returns false if the device path could not be used (more than
// 4096 colours, timeout): the caller then runs the host algorithm instead, loudly.
inline bool color_firstfit_device(long long n, const std::vector<long long>& keep, const std::vector<long long>& cptr, const uvector<int>& crow,
                                  const uvector<int>& cpos, const std::vector<long long>& rowptr, const uvector<int>& col, std::vector<int>& colors,
                                  hipStream_t st) {
    const long long nKeep = (long long)keep.size();
    const bool dbg = getenv("DAS_DEBUG_TIMING") != nullptr;
    double tq = wall_seconds();
    auto lap = [&](const char* what) { if (dbg) { double t2 = wall_seconds(); fprintf(stderr, "[dafoam_amd]     device colouring: %s %.2f s\n", what, t2 - tq); tq = t2; } };
    // crow holds row ids of the full pattern: translate to kept-row positions (net ids)
    std::vector<int> posOfRow;
    {
        long long nrows = (long long)rowptr.size() - 1;
        posOfRow.assign(nrows, -1);
        for (long long q = 0; q < nKeep; q++) posOfRow[keep[q]] = (int)q;
    }
    uvector<int> crowK(crow.size());
#pragma omp parallel for schedule(static) num_threads(das::host_threads())
    for (long long q = 0; q < (long long)crow.size(); q++) crowK[q] = posOfRow[crow[q]];
    // groups: maximal runs of consecutive columns with identical kept-row lists (the xyz components of a cell's U)
    std::vector<unsigned char> isStart(n, 1);
#pragma omp parallel for schedule(static) num_threads(das::host_threads())
    for (long long j = 1; j < n; j++) {
        const long long len = cptr[j + 1] - cptr[j];
        isStart[j] = !(cptr[j] - cptr[j - 1] == len && std::equal(crow.begin() + cptr[j], crow.begin() + cptr[j + 1], crow.begin() + cptr[j - 1]));
    }
    const std::vector<long long> gstart = color_groups_from_flags(n, isStart);
    lap("host preparation");
    DevBuf<long long> d_cptr;
    DevBuf<int> d_crow, d_cpos;
    d_cptr.upload(cptr);
    d_crow.upload(crowK.data(), crowK.size()); d_cpos.upload(cpos.data(), cpos.size());
    lap("upload");
    (void)col;
    const int rc = color_firstfit_run(n, nKeep, gstart, d_cptr.p, d_crow.p, d_cpos.p, colors, st);
    lap("kernel");
    return rc == 1;
}

// =====================================================================================================================
// Speculative distance-2 colouring over NET BITMAPS (round 3; amd.coloringAlgorithm "speculative", the default on a device).
//
// The data-flow first-fit above is exact (the serial colours) but it is work-bound: every column gathers the colours of its
// whole distance-2 neighbourhood WITH multiplicity - ~63 kept rows x ~275 columns = 17 k gathers per column, 2e11 at 2 M cells,
// 9.8 s.  Here every kept row ("net": one per cell, its pRes row) carries a bitmap of the colours its columns use (W 64-bit
// words); a column reads its forbidden set as the OR of the bitmaps of its ~63 nets - 63 coalesced 64-byte lines instead of
// 17 k scattered words - and the rounds are the classic speculate / detect / retry scheme (Gebremedhin-Manne; net-based
// detection as in Tas, Kaya, Saule 2017), made deterministic by index priority:
//   assign    every uncoloured column picks the first colour not in its forbidden set, starting the search at a hashed
//             offset inside [0, S) (S ~ 1.2 x the longest net: spreads the simultaneous picks), beyond S only if [0, S) is full
//   conflict  every net looks at its columns: of two that hold the same colour the committed one, else the lower index, wins
//   commit    winners become final, losers go back to "uncoloured";  netbits: the bitmaps are rebuilt from the final colours
// Every round commits at least the lowest-index contender of every colour class, so it terminates; measured rounds /
// colours / time: profiles/README.md.  The result depends on nothing but the pattern (no race decides anything).
// Reference contract: any valid colouring (DAColoring::validateColoring, DAColoring.C:931-1037); validity is checked by the
// same net kernel at the end.
struct SpecView {
    long long n, nNets;
    const long long* cptr;  // n+1: nets of a column
    const int* crow;
    const long long* krp;   // nNets+1: columns of a net
    const int* kcol;
    int W;                  // bitmap words per net
    unsigned long long* F;  // nNets * W
    int* colors;            // final colours, -1 = none
    int* tent;              // tentative colour of this round, -1 = none
    unsigned char* lose;
    unsigned* ctrl;         // [0] columns still uncoloured after commit, [1] overflow (no free colour in 64 W), [2] invalid (check)
};
__device__ __forceinline__ unsigned spec_hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// 8 lanes per column (one bitmap word each when W == 8; W > 8: strided)
__global__ __launch_bounds__(256) void k_spec_assign(SpecView P, int S, unsigned round) {
    const long long j = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int l8 = threadIdx.x & 7;
    if (j >= P.n) return;
    if (P.colors[j] >= 0) return;
    const int W = P.W;
    unsigned h = spec_hash((unsigned)j * 2654435761u + round * 40503u);
    const int start = (int)(h % (unsigned)S);
    // forbidden words owned by this lane: w = l8, l8 + 8, ...
    int best = 1 << 30, bestWrap = 1 << 30;  // first free colour >= start (in [0,S) first, then anywhere), first free colour < start
    for (int w = l8; w < W; w += 8) {
        unsigned long long f = 0ull;
        for (long long q = P.cptr[j]; q < P.cptr[j + 1]; q++) f |= P.F[(long long)P.crow[q] * W + w];
        unsigned long long freeb = ~f;
        const int base = w * 64;
        // candidates at or after start
        unsigned long long hi = freeb;
        if (base + 64 <= start) hi = 0ull;
        else if (base < start) hi &= ~0ull << (start - base);
        if (hi) best = min(best, base + __builtin_ctzll(hi));
        unsigned long long lo = freeb;
        if (base >= start) lo = 0ull;
        else if (base + 64 > start) lo &= (1ull << (start - base)) - 1ull;
        if (lo) bestWrap = min(bestWrap, base + __builtin_ctzll(lo));
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        best = min(best, __shfl_xor(best, o, 8));
        bestWrap = min(bestWrap, __shfl_xor(bestWrap, o, 8));
    }
    // inside [0, S): first free at/after start, else wrapped; if [0, S) is full: the first free colour beyond S
    int c = (best < S) ? best : (bestWrap < (1 << 30) ? bestWrap : best);
    if (l8 == 0) {
        if (c >= 64 * W) { P.ctrl[1] = 1u; c = -1; }
        P.tent[j] = c;
    }
}
// one wave per net: of the columns that hold the same colour only one survives (final ones first, then the lowest index)
__global__ __launch_bounds__(256) void k_spec_conflict(SpecView P, int check) {
    extern __shared__ int sh_min[];  // 4 waves x 64 W entries
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + wave;
    const int NC = 64 * P.W;
    int* mn = sh_min + wave * NC;
    for (int c = lane; c < NC; c += 64) mn[c] = 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    if (r >= P.nNets) return;
    const long long k0 = P.krp[r], k1 = P.krp[r + 1];
    for (long long k = k0 + lane; k < k1; k += 64) {
        const int j = P.kcol[k];
        const int cf = P.colors[j];
        if (cf >= 0) {
            const int old = atomicMin(&mn[cf], -1 - j);  // final colours: negative keys (always beat tentative ones)
            if (check && old < 0 && old != -1 - j) P.ctrl[2] = 1u;  // two FINAL columns of one net share a colour
        } else {
            if (check) P.ctrl[2] = 1u;  // an uncoloured column
            const int ct = P.tent[j];
            if (ct >= 0) atomicMin(&mn[ct], j);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (check) return;
    for (long long k = k0 + lane; k < k1; k += 64) {
        const int j = P.kcol[k];
        if (P.colors[j] >= 0) continue;
        const int ct = P.tent[j];
        if (ct >= 0 && mn[ct] != j) P.lose[j] = 1;
    }
}
__global__ void k_spec_commit(SpecView P) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.n || P.colors[j] >= 0) return;
    const int ct = P.tent[j];
    const bool won = ct >= 0 && !P.lose[j];
    if (won) P.colors[j] = ct;
    P.tent[j] = -1;
    P.lose[j] = 0;
    // columns still to colour: one atomic per wave (a counter hit by every thread retires one add per ~17 ns)
    const unsigned long long m = __ballot(!won);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(&P.ctrl[0], (unsigned)__builtin_popcountll(m));
}
// one wave per net: bitmap of the final colours of its columns
__global__ __launch_bounds__(256) void k_spec_netbits(SpecView P) {
    extern __shared__ unsigned long long sh_bits[];  // 4 waves x W words
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + wave;
    unsigned long long* bits = sh_bits + wave * P.W;
    for (int w = lane; w < P.W; w += 64) bits[w] = 0ull;
    __builtin_amdgcn_wave_barrier();
    if (r >= P.nNets) return;
    for (long long k = P.krp[r] + lane; k < P.krp[r + 1]; k += 64) {
        const int c = P.colors[P.kcol[k]];
        if (c >= 0) atomicOr(&bits[c >> 6], 1ull << (c & 63));
    }
    __builtin_amdgcn_wave_barrier();
    for (int w = lane; w < P.W; w += 64) P.F[r * P.W + w] = bits[w];
}

// the rounds of the speculative colouring on device arrays: column -> nets (cptr / crow) and net -> columns (krp / kcol)
inline bool color_speculative_run(long long n, long long nKeep, long long maxNet, const long long* d_cptr, const int* d_crow, const long long* d_krp,
                                  const int* d_kcol, std::vector<int>& colors, hipStream_t st, int* roundsOut = nullptr) {
    const bool dbg = getenv("DAS_DEBUG_TIMING") != nullptr;
    DevBuf<int> d_colors(n), d_tent(n);
    DevBuf<unsigned char> d_lose(n);
    DevBuf<unsigned> d_ctrl(4);
    // the spread of the first picks: a little above the longest net (a lower bound of the colour count)
    int S = (int)std::max<long long>(8, maxNet + maxNet / 4);
    if (const char* e = getenv("DAS_COLOR_SPREAD")) S = std::max(1, atoi(e));
    for (int W = (int)std::max<long long>(8, ((long long)(2 * S) + 63) / 64); W <= 1024; W *= 2) {
        DevBuf<unsigned long long> d_F((size_t)nKeep * W);
        DAS_HIP(hipMemsetAsync(d_F.p, 0, (size_t)nKeep * W * sizeof(unsigned long long), st));
        DAS_HIP(hipMemsetAsync(d_colors.p, 0xff, n * sizeof(int), st));
        DAS_HIP(hipMemsetAsync(d_tent.p, 0xff, n * sizeof(int), st));
        DAS_HIP(hipMemsetAsync(d_lose.p, 0, n, st));
        SpecView V{n, nKeep, d_cptr, d_crow, d_krp, d_kcol, W, d_F.p, d_colors.p, d_tent.p, d_lose.p, d_ctrl.p};
        const unsigned gNet = (unsigned)((nKeep + 3) / 4);
        bool overflow = false;
        unsigned left = 1u;
        int round = 0;
        for (; left != 0u && round < 100000; round++) {
            DAS_HIP(hipMemsetAsync(d_ctrl.p, 0, 4 * sizeof(unsigned), st));
            hipLaunchKernelGGL(k_spec_assign, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, V, S, (unsigned)round);
            hipLaunchKernelGGL(k_spec_conflict, dim3(gNet), dim3(256), (size_t)4 * 64 * W * sizeof(int), st, V, 0);
            hipLaunchKernelGGL(k_spec_commit, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, V);
            hipLaunchKernelGGL(k_spec_netbits, dim3(gNet), dim3(256), (size_t)4 * W * sizeof(unsigned long long), st, V);
            unsigned ctrl[4] = {0, 0, 0, 0};
            DAS_HIP(hipMemcpyAsync(ctrl, d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost, st));
            DAS_HIP(hipStreamSynchronize(st));
            left = ctrl[0];
            if (ctrl[1]) { overflow = true; break; }
        }
        if (overflow) continue;  // more colours than the bitmaps hold: twice the words
        if (left != 0u) return false;
        // validity: every net holds pairwise different final colours
        DAS_HIP(hipMemsetAsync(d_ctrl.p, 0, 4 * sizeof(unsigned), st));
        hipLaunchKernelGGL(k_spec_conflict, dim3(gNet), dim3(256), (size_t)4 * 64 * W * sizeof(int), st, V, 1);
        unsigned ctrl[4] = {0, 0, 0, 0};
        DAS_HIP(hipMemcpyAsync(ctrl, d_ctrl.p, sizeof(ctrl), hipMemcpyDeviceToHost, st));
        DAS_HIP(hipStreamSynchronize(st));
        if (ctrl[2]) return false;
        colors.resize(n);
        d_colors.download(colors.data(), n);
        if (roundsOut) *roundsOut = round;
        if (dbg) fprintf(stderr, "[dafoam_amd]     speculative colouring: %d rounds, spread %d, %d bitmap words per net\n", round, S, W);
        return true;
    }
    return false;
}
// speculative colouring on the device; same inputs as color_firstfit_device.  Returns false if it could not be used.
inline bool color_speculative_device(long long n, const std::vector<long long>& keep, const std::vector<long long>& cptr, const uvector<int>& crow,
                                     const std::vector<long long>& rowptr, const uvector<int>& col, std::vector<int>& colors, hipStream_t st,
                                     int* roundsOut = nullptr) {
    const long long nKeep = (long long)keep.size();
    const bool dbg = getenv("DAS_DEBUG_TIMING") != nullptr;
    double tq = wall_seconds();
    auto lap = [&](const char* what) { if (dbg) { double t2 = wall_seconds(); fprintf(stderr, "[dafoam_amd]     speculative colouring: %s %.2f s\n", what, t2 - tq); tq = t2; } };
    std::vector<long long> krp(nKeep + 1, 0);
    long long maxNet = 0;
    for (long long q = 0; q < nKeep; q++) {
        const long long len = rowptr[keep[q] + 1] - rowptr[keep[q]];
        krp[q + 1] = krp[q] + len;
        maxNet = std::max(maxNet, len);
    }
    uvector<int> kcol(krp[nKeep]);
#pragma omp parallel for schedule(static) num_threads(das::host_threads())
    for (long long q = 0; q < nKeep; q++) std::copy(col.begin() + rowptr[keep[q]], col.begin() + rowptr[keep[q] + 1], kcol.begin() + krp[q]);
    std::vector<int> posOfRow((long long)rowptr.size() - 1, -1);
    for (long long q = 0; q < nKeep; q++) posOfRow[keep[q]] = (int)q;
    uvector<int> crowK(crow.size());
#pragma omp parallel for schedule(static) num_threads(das::host_threads())
    for (long long q = 0; q < (long long)crow.size(); q++) crowK[q] = posOfRow[crow[q]];
    lap("host preparation");
    DevBuf<long long> d_cptr, d_krp;
    DevBuf<int> d_crow, d_kcol;
    d_cptr.upload(cptr); d_krp.upload(krp);
    d_crow.upload(crowK.data(), crowK.size()); d_kcol.upload(kcol.data(), kcol.size());
    lap("upload");
    const bool ok = color_speculative_run(n, nKeep, maxNet, d_cptr.p, d_crow.p, d_krp.p, d_kcol.p, colors, st, roundsOut);
    lap("kernels");
    return ok;
}

}  // namespace das
