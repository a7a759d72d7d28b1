// Mesh-sensitivity product  product = [dOutput/dX]^T seeds  for ALL mesh points at once (3 nPoints entries), on the device.
//
// Reference: calcJacTVecProduct(inputType "volCoord" -> residual | function), DASolver.C:1690-1839 with DAInputVolCoord.C:
// one reverse sweep of the CoDiPack tape from the seeded outputs back to the point coordinates.  There is no tape here.  The
// product is assembled from coloured central differences that never leave the GPU:
//   * the cells whose residual rows can feel point p (its influence set, build_point_influence in das_mesh.cpp) are disjoint
//     for two points of one colour, so ONE pass moves every point of a colour by +h_p (then -h_p) along one axis;
//   * per pass: k_move_points -> the three metric passes of das_geom.hpp as kernels (k_geom_face / cell / weights, 21 M
//     entity updates at 2 M cells) -> the ordinary residual evaluation (or k_grad + k_fn_value, the per-face summands, for an objective);
//   * t_i = sum over the rows of cell i of seeds_r (R+_r - R-_r)  (k_vc_rows: cell rows + the phi rows of the faces the cell
//     owns), then one wavefront per moved point gathers  product[3 p + axis] = sum_{i in influence(p)} t_i / (2 h_p)
//     (k_vc_gather) - fixed summation order, no atomics: the product is bit-reproducible.
// Cost: 2 x 3 x nColours metric + residual passes (a few hundred colours on a hex mesh); accuracy: central differences with
// a step of 1e-4 of the smallest adjacent cell thickness (truncation ~1e-8, rounding ~1e-12 relative).
#pragma once
#include "das_geom.hpp"
#include "das_kernels.hpp"

namespace das {

template <class S>
__global__ __launch_bounds__(256) void k_geom_face(GeomTopo t, const S* __restrict__ P, FaceGeomT<S>* fg) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < t.nF) geom_face<S>(f, t, P, fg[f]);
}
template <class S>
__global__ __launch_bounds__(256) void k_geom_cell(GeomTopo t, const FaceGeomT<S>* __restrict__ fg, CellGeomT<S>* cg, int* bad) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < t.nC && !geom_cell<S>(c, t, fg, cg[c])) atomicExch(bad, 1);
}
template <class S>
__global__ __launch_bounds__(256) void k_geom_weights(GeomTopo t, const CellGeomT<S>* __restrict__ cg, FaceGeomT<S>* fg) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < t.nF) geom_weights<S>(f, t, cg, fg, fg[f]);
}
// ---- exact mode (amd.volCoordMode "dual"): the point coordinates carry a unit tangent, the metrics and the residual follow as
//      Dual<1>: ONE pass per colour and axis gives d(rows)/d(point coordinate) without a step size - and without the errors a
//      difference makes where a limiter switch lies inside the step (tests/test_host_cpu.py::test_dual_number_metrics_...)
typedef Dual<1> VD;
// XD = (X0, 0) for all coordinates
__global__ void k_points_lift(long long n3, const double* __restrict__ X0, VD* __restrict__ XD) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) XD[i] = VD(X0[i]);
}
// tangent of coordinate `axis` of the points of one colour := t (1 before the pass, 0 after it)
__global__ void k_points_seed(int np, const int* __restrict__ pts, int axis, double t, VD* __restrict__ XD) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < np) XD[3LL * pts[k] + axis].d[0] = t;
}
// the frozen wall distance as a Dual without tangent (cg records are rewritten by k_geom_cell except for y)
__global__ void k_cell_y(int nC, const CellGeom* __restrict__ cg, CellGeomT<VD>* __restrict__ cgD) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nC) cgD[c].y = VD(cg[c].y);
}
// t_i = sum over the rows of cell i of seeds_r dR_r (the tangent part of the dual residual)

// X[3 p + axis] = X0[3 p + axis] + sgn h_p for the points of one colour (sgn = 0 restores them)
__global__ void k_move_points(int np, const int* __restrict__ pts, const double* __restrict__ h, double sgn, int axis, const double* __restrict__ X0,
                              double* __restrict__ X) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= np) return;
    const long long p = pts[k];
    X[3 * p + axis] = X0[3 * p + axis] + sgn * h[p];
}

struct RowLayout {  // DAIndex "state" ordering: block b holds 3 N (vector), N (scalar) or nF (face) rows from off[b]
    int nb;
    long long off[8];
    int kind[8];
};
// t_i = sum over the rows of cell i of seeds_r (Rp_r - Rm_r): its cell-centred rows and the face rows of the faces it owns
__global__ __launch_bounds__(256) void k_vc_rows(DevMesh m, RowLayout L, const double* __restrict__ seeds, const double* __restrict__ Rp,
                                                 const double* __restrict__ Rm, double* __restrict__ tc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m.nC) return;
    double acc = 0.0;
    for (int b = 0; b < L.nb; b++) {
        if (L.kind[b] == KIND_VEC) {
            for (int q = 0; q < 3; q++) { const long long r = L.off[b] + 3LL * c + q; acc += seeds[r] * (Rp[r] - Rm[r]); }
        } else if (L.kind[b] == KIND_SCL) {
            const long long r = L.off[b] + c;
            acc += seeds[r] * (Rp[r] - Rm[r]);
        } else {
            for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
                const int fe = m.cf_face[s];
                if (fe < 0) continue;  // the cell is the face's neighbour: the row belongs to the owner
                const long long r = L.off[b] + fe;
                acc += seeds[r] * (Rp[r] - Rm[r]);
            }
        }
    }
    tc[c] = acc;
}
// exact mode: t_i = sum over the rows of cell i of seeds_r dR_r (the tangent part of the dual residual)
__global__ __launch_bounds__(256) void k_vc_rows_dual(DevMesh m, RowLayout L, const double* __restrict__ seeds, const VD* __restrict__ Rd,
                                                      double* __restrict__ tc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m.nC) return;
    double acc = 0.0;
    for (int b = 0; b < L.nb; b++) {
        if (L.kind[b] == KIND_VEC) {
            for (int q = 0; q < 3; q++) { const long long r = L.off[b] + 3LL * c + q; acc += seeds[r] * Rd[r].d[0]; }
        } else if (L.kind[b] == KIND_SCL) {
            const long long r = L.off[b] + c;
            acc += seeds[r] * Rd[r].d[0];
        } else {
            for (int s = m.cf_ptr[c]; s < m.cf_ptr[c + 1]; s++) {
                const int fe = m.cf_face[s];
                if (fe < 0) continue;
                const long long r = L.off[b] + fe;
                acc += seeds[r] * Rd[r].d[0];
            }
        }
    }
    tc[c] = acc;
}
// exact mode, objectives: the tangent of the per-face summand w_k q_k (force, mass flow: constant weights; moment: the arm
// axis x (Cf - center) follows the dual face centre; area averages: the linearised functional (cN q + cA) |Sf| as in k_fn_area_avg)
template <bool RHO>
__global__ __launch_bounds__(256) void k_fn_dual(DevMeshT<VD> m, ResParams prm, const VD* __restrict__ W, const VD* nut, const VD* gU, FaceFnView fn,
                                                 int isMoment, double a0, double a1, double a2, double c0, double c1, double c2, int areaAvg, double cN0,
                                                 double cN1, double cA0, double cA1, double* __restrict__ fv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= fn.nf) return;
    const int f = fn.faces[k];
    VD dir[3] = {VD(0.0), VD(0.0), VD(0.0)};
    if (isMoment) {
        const FaceGeomT<VD>& g = m.fg[f];
        const VD r0 = g.Cf[0] - c0, r1 = g.Cf[1] - c1, r2 = g.Cf[2] - c2;
        dir[0] = a1 * r2 - a2 * r1; dir[1] = a2 * r0 - a0 * r2; dir[2] = a0 * r1 - a1 * r0;
    } else if (fn.dir) {
        dir[0] = VD(fn.dir[3 * k]); dir[1] = VD(fn.dir[3 * k + 1]); dir[2] = VD(fn.dir[3 * k + 2]);
    }
    const VD q = body_facefn<VD, RHO>(f, m, prm, W, nut, gU, fn.kind, dir, fn.gammaFn, fn.RFn);
    VD v;
    if (areaAvg) v = (fn.group[k] ? (cN1 * q + cA1) : (cN0 * q + cA0)) * m.fg[f].magSf;
    else v = fn.w[k] * q;
    fv[k] = v.d[0];
}
// t_i = seed x sum over the function faces of cell i of the tangents
__global__ __launch_bounds__(256) void k_vc_fn_cells_dual(int nC, const int* __restrict__ cfPtr, const int* __restrict__ cfIdx, double seed,
                                                          const double* __restrict__ fvd, double* __restrict__ tc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nC) return;
    double acc = 0.0;
    for (int q = cfPtr[c]; q < cfPtr[c + 1]; q++) acc += fvd[cfIdx[q]];
    tc[c] = seed * acc;
}
// moment functions: (r x F) . axis = F . (axis x r), r = Cf - center - the per-face direction follows the moved face centres
__global__ void k_fn_moment_dir(int nf, const int* __restrict__ faces, const FaceGeom* __restrict__ fg, double a0, double a1, double a2, double c0, double c1,
                                double c2, double* __restrict__ dir) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nf) return;
    const FaceGeom& g = fg[faces[k]];
    const double r0 = g.Cf[0] - c0, r1 = g.Cf[1] - c1, r2 = g.Cf[2] - c2;
    dir[3 * k] = a1 * r2 - a2 * r1;
    dir[3 * k + 1] = a2 * r0 - a0 * r2;
    dir[3 * k + 2] = a0 * r1 - a1 * r0;
}
// area-averaged functions (totalPressure: F = scale N / A; totalTemperatureRatio: F = (N1 / A1) / (N0 / A0) with N_g = sum |Sf| q,
// A_g = sum |Sf| over the faces of group g): the host-built weights |Sf| / A carry the metrics, so the product differentiates the
// LINEARISED functional  fv_k = cN[g] |Sf_k| q_k + cA[g] |Sf_k|  (cN = dF/dN_g, cA = dF/dA_g at the unperturbed mesh) on the
// perturbed metrics instead
template <bool RHO>
__global__ __launch_bounds__(256) void k_fn_area_avg(DevMesh m, ResParams prm, const double* __restrict__ W, const double* nut, const double* gU,
                                                     FaceFnView fn, double cN0, double cN1, double cA0, double cA1, double* __restrict__ fv) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= fn.nf) return;
    double dir[3] = {0.0, 0.0, 0.0};
    const int f = fn.faces[k];
    const double q = body_facefn<double, RHO>(f, m, prm, W, nut, gU, fn.kind, dir, fn.gammaFn, fn.RFn);
    const double a = m.fg[f].magSf;
    fv[k] = fn.group[k] ? (cN1 * q + cA1) * a : (cN0 * q + cA0) * a;
}
// t_i = seed x sum over the function faces of cell i of (fv+ - fv-)   (cellFn: CSR cell -> function-face slots)
__global__ __launch_bounds__(256) void k_vc_fn_cells(int nC, const int* __restrict__ cfPtr, const int* __restrict__ cfIdx, double seed,
                                                     const double* __restrict__ fvp, const double* __restrict__ fvm, double* __restrict__ tc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nC) return;
    double acc = 0.0;
    for (int q = cfPtr[c]; q < cfPtr[c + 1]; q++) acc += fvp[cfIdx[q]] - fvm[cfIdx[q]];
    tc[c] = seed * acc;
}
// one wavefront per moved point: product[3 p + axis] = sum over its influence set of t_i / (2 h_p)
// (h == nullptr: exact mode, the sums are the derivatives themselves)
__global__ __launch_bounds__(256) void k_vc_gather(int np, const int* __restrict__ pts, const long long* __restrict__ ptr, const int* __restrict__ cells,
                                                   const double* __restrict__ tc, const double* __restrict__ h, int axis, double* __restrict__ product) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= np) return;
    const long long p = pts[k];
    double acc = 0.0;
    for (long long q = ptr[p] + lane; q < ptr[p + 1]; q += 64) acc += tc[cells[q]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) product[3 * p + axis] = h ? acc / (2.0 * h[p]) : acc;
}

}  // namespace das
