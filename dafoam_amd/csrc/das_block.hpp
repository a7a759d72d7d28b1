// Block (multi right-hand-side) Krylov kernels: the tall-skinny fp64 GEMMs of the block orthogonalisation on the matrix
// cores (v_mfma_f64_16x16x4_f64) and the sparse matrix x s-vector product that streams dRdW^T once for s adjoints.
//
// Reference behaviour being improved on: the reference solves one adjoint per objective function, one after the other
// (dafoam/mphys/mphys_dafoam.py:478-481 loops over the functions); here the CD and CL adjoints (s = 2..8 right-hand
// sides) advance together through ONE block GMRES: A is read once per iteration for all s vectors, the inner products
// V^T W ((j+1)s x s) and the update W -= V H are GEMMs with one long dimension (n) - bandwidth-bound, but shaped for MFMA
// (BASELINE.json north_star: "MFMA only for the tall-skinny orthogonalisation block").
//
// Layout: every basis / work vector is contiguous (column-major blocks, leading dimension n).  f64 MFMA fragment maps
// (cdna_hip_programming.md section 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C/D: col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma once
#include "das_common.hpp"

namespace das {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

#define TSG_TILES 4      // basis-vector tiles (of 16) per wave in the TN product: W is re-read once per 64 basis vectors
#define TSG_WAVES 4

// partial[(chunk * Kpad + i) * 16 + r] = sum over the chunk's rows of V_i[row] * W_r[row]   (i < K, r < s <= 16)
// grid = (nChunks / TSG_WAVES, ceil(K / (16 TSG_TILES))), one wave per (row chunk, group of 64 basis vectors).
// Row mapping (round 3; the first version read 4 consecutive rows of 16 vectors per instruction - 32-byte segments, and ran only
// 256 waves: 4 x off the bandwidth in the profile): a step covers 16 rows; lane (vector li, quarter lk) loads rows
// base + 4 lk .. + 3 of ITS vector as two 16-byte loads, i.e. every vector contributes one full 128-byte line per step, and the
// four MFMAs of the step take k = lk with row base + 4 lk + t - the sum over rows does not care in which order they enter, as long
// as the A (V) and B (W) fragments use the same mapping.
__global__ __launch_bounds__(64 * TSG_WAVES) void k_tsgemm_tn(long long n, int K, int s, const double* __restrict__ V, long long ldv,
                                                              const double* __restrict__ W, long long ldw, long long rowsPerChunk, int Kpad,
                                                              double* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const long long chunk = (long long)blockIdx.x * TSG_WAVES + (threadIdx.x >> 6);
    const int i0 = blockIdx.y * 16 * TSG_TILES;
    const long long r0 = chunk * rowsPerChunk, r1 = min(n, r0 + rowsPerChunk);  // rowsPerChunk is a multiple of 16, r0 16-byte aligned
    const int li = lane & 15, lk = lane >> 4;
    mfma_d4 acc[TSG_TILES];
#pragma unroll
    for (int t = 0; t < TSG_TILES; t++) acc[t] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    const double* wp = W + (long long)min(li, s - 1) * ldw;
    const bool wact = li < s;
    const double* vp[TSG_TILES];
    bool vact[TSG_TILES];
#pragma unroll
    for (int t = 0; t < TSG_TILES; t++) {
        const int i = i0 + 16 * t + li;
        vact[t] = i < K;
        vp[t] = V + (long long)min(i, K - 1) * ldv;
    }
    const bool al = ((ldv | ldw) & 1) == 0;  // even leading dimensions: 16-byte aligned row quadruples
    for (long long base = r0; base < r1; base += 16) {  // wave-uniform trip count (MFMA needs the whole wave)
        const long long row = base + 4 * lk;
        double bq[4], aq[TSG_TILES][4];
        if (row + 3 < r1 && al) {
            const double2 b01 = *reinterpret_cast<const double2*>(wp + row), b23 = *reinterpret_cast<const double2*>(wp + row + 2);
            bq[0] = wact ? b01.x : 0.0; bq[1] = wact ? b01.y : 0.0; bq[2] = wact ? b23.x : 0.0; bq[3] = wact ? b23.y : 0.0;
#pragma unroll
            for (int t = 0; t < TSG_TILES; t++) {
                const double2 a01 = *reinterpret_cast<const double2*>(vp[t] + row), a23 = *reinterpret_cast<const double2*>(vp[t] + row + 2);
                aq[t][0] = vact[t] ? a01.x : 0.0; aq[t][1] = vact[t] ? a01.y : 0.0; aq[t][2] = vact[t] ? a23.x : 0.0; aq[t][3] = vact[t] ? a23.y : 0.0;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = row + u < r1;
                bq[u] = (wact && in) ? wp[row + u] : 0.0;
#pragma unroll
                for (int t = 0; t < TSG_TILES; t++) aq[t][u] = (vact[t] && in) ? vp[t][row + u] : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < TSG_TILES; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[t][u], bq[u], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < TSG_TILES; t++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int i = i0 + 16 * t + lk + 4 * v;  // row of the C tile
            if (i < Kpad) partial[(chunk * Kpad + i) * 16 + li] = acc[t][v];
        }
}
// C[i * s + r] = sum over chunks of partial: one wavefront per entry (lanes over the chunks, fixed order: deterministic)
__global__ __launch_bounds__(256) void k_tsgemm_reduce(int K, int s, int Kpad, long long nChunks, const double* __restrict__ partial, double* __restrict__ C) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= K * s) return;
    const int i = idx / s, r = idx % s;
    double acc = 0.0;
    for (long long c = lane; c < nChunks; c += 64) acc += partial[(c * Kpad + i) * 16 + r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) C[idx] = acc;
}

// W_r[row] -= sum_i V_i[row] * C[i * s + r]: one wave per 64 rows (4 row tiles of 16), K loop in steps of 4
__global__ __launch_bounds__(256) void k_tsgemm_nn_sub(long long n, int K, int s, const double* __restrict__ V, long long ldv,
                                                       const double* __restrict__ C, double* __restrict__ W, long long ldw) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (row0 >= n) return;
    const int lm = lane & 15, lk = lane >> 4;
    mfma_d4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (mfma_d4){0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < K; i += 4) {
        const int ik = i + lk;
        const double b = (ik < K && lm < s) ? C[ik * s + lm] : 0.0;  // B[k][j = lm]
        const double* vi = V + (long long)min(ik, K - 1) * ldv;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const long long row = row0 + 16 * t + lm;
            const double a = (ik < K && row < n) ? vi[row] : 0.0;       // A[m = lm][k]
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
    if (lm < s) {
        double* wr = W + (long long)lm * ldw;
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const long long row = row0 + 16 * t + lk + 4 * v;
                if (row < n) wr[row] -= acc[t][v];
            }
    }
}

// column-major block (s vectors, ld n) -> row-major n x S (zero padded): the gather layout of the sparse product
template <int S>
__global__ void k_block_to_rows(long long n, int s, const double* __restrict__ X, long long ldx, double* __restrict__ Xr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int r = 0; r < S; r++) Xr[i * S + r] = r < s ? X[(long long)r * ldx + i] : 0.0;
}
// Y_r = A X_r for r < s: the matrix entries are read ONCE for all s vectors (16 lanes per row)
template <int S>
__global__ __launch_bounds__(256) void k_spmm_wave(long long n, int s, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                   const double* __restrict__ v, const double* __restrict__ Xr, double* __restrict__ Y, long long ldy) {
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int lane = threadIdx.x & 15;
    if (row >= n) return;
    double acc[S];
#pragma unroll
    for (int r = 0; r < S; r++) acc[r] = 0.0;
    constexpr int U = S <= 4 ? 4 : 2;  // entries in flight per lane (value, column, S gathered doubles each)
    const long long b = rp[row], e = rp[row + 1];
    for (long long k = b + lane; k - lane < e; k += U * 16) {
        double a[U];
        int c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long kk = k + u * 16;
            const long long kc = kk < e ? kk : e - 1;  // clamped + masked: the tail keeps all U loads in flight
            a[u] = kk < e ? v[kc] : 0.0;
            c[u] = ci[kc];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const double* x = Xr + (long long)c[u] * S;
#pragma unroll
            for (int r = 0; r < S; r++) acc[r] += a[u] * x[r];
        }
    }
#pragma unroll
    for (int r = 0; r < S; r++) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc[r] += __shfl_down(acc[r], o, 16);
    }
    if (lane == 0)
        for (int r = 0; r < s; r++) Y[(long long)r * ldy + row] = acc[r];
}
// W <- W T for a small s x s matrix T (row-major, e.g. the inverse Cholesky factor): every row of the block in place
__global__ void k_block_right_mult(long long n, int s, double* __restrict__ W, long long ldw, const double* __restrict__ T) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double w[8], o[8];
    for (int r = 0; r < s; r++) w[r] = W[(long long)r * ldw + i];
    for (int c = 0; c < s; c++) {
        double a = 0.0;
        for (int r = 0; r < s; r++) a += w[r] * T[r * s + c];
        o[c] = a;
    }
    for (int c = 0; c < s; c++) W[(long long)c * ldw + i] = o[c];
}
// Y = sum_i V_i C[i * s + r] (block linear combination; used for the solution update)
__global__ void k_block_lincomb(long long n, int K, int s, const double* __restrict__ V, long long ldv, const double* __restrict__ C,
                                double* __restrict__ Y, long long ldy) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < K; k++) {
        const double vk = V[(long long)k * ldv + i];
        for (int r = 0; r < s; r++) acc[r] += vk * C[k * s + r];
    }
    for (int r = 0; r < s; r++) Y[(long long)r * ldy + i] = acc[r];
}

}  // namespace das
