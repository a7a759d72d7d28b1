// Forward-mode dual numbers for the coloured Jacobian assembly.
//
// The reference obtains exact dR/dW^T products from a CoDiPack reverse tape
// (reference src/adjoint/DASolver/DASolver.C:1364-1441).  A tape has no sensible GPU
// counterpart; instead every residual kernel is templated on the scalar type and the
// same graph colouring that drives the reference's finite-difference Jacobian
// (DAPartDeriv.C:350-473) carries dual-number seeds: one pass = value + K tangents.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace das {

template <int K>
struct Dual {
    double v;
    double d[K];
    __host__ __device__ Dual() {}
    __host__ __device__ Dual(double a) : v(a) {
#pragma unroll
        for (int k = 0; k < K; k++) d[k] = 0.0;
    }
};

#define DAS_HD __host__ __device__ __forceinline__

DAS_HD double val(double a) { return a; }
template <int K>
DAS_HD double val(const Dual<K>& a) { return a.v; }

template <int K>
DAS_HD Dual<K> operator+(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = a.d[k] + b.d[k];
    return r;
}
template <int K>
DAS_HD Dual<K> operator-(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = a.d[k] - b.d[k];
    return r;
}
template <int K>
DAS_HD Dual<K> operator-(const Dual<K>& a) {
    Dual<K> r;
    r.v = -a.v;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = -a.d[k];
    return r;
}
template <int K>
DAS_HD Dual<K> operator*(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = a.v * b.d[k] + a.d[k] * b.v;
    return r;
}
template <int K>
DAS_HD Dual<K> operator/(const Dual<K>& a, const Dual<K>& b) {
    Dual<K> r;
    double ib = 1.0 / b.v;
    r.v = a.v * ib;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = (a.d[k] - r.v * b.d[k]) * ib;
    return r;
}
// mixed with double
template <int K> DAS_HD Dual<K> operator+(const Dual<K>& a, double b) { Dual<K> r = a; r.v += b; return r; }
template <int K> DAS_HD Dual<K> operator+(double b, const Dual<K>& a) { Dual<K> r = a; r.v += b; return r; }
template <int K> DAS_HD Dual<K> operator-(const Dual<K>& a, double b) { Dual<K> r = a; r.v -= b; return r; }
template <int K> DAS_HD Dual<K> operator-(double b, const Dual<K>& a) { Dual<K> r = -a; r.v += b; return r; }
template <int K>
DAS_HD Dual<K> operator*(const Dual<K>& a, double b) {
    Dual<K> r;
    r.v = a.v * b;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = a.d[k] * b;
    return r;
}
template <int K> DAS_HD Dual<K> operator*(double b, const Dual<K>& a) { return a * b; }
template <int K> DAS_HD Dual<K> operator/(const Dual<K>& a, double b) { return a * (1.0 / b); }
template <int K>
DAS_HD Dual<K> operator/(double a, const Dual<K>& b) {
    Dual<K> r;
    double ib = 1.0 / b.v;
    r.v = a * ib;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = -r.v * b.d[k] * ib;
    return r;
}
template <int K> DAS_HD Dual<K>& operator+=(Dual<K>& a, const Dual<K>& b) { a = a + b; return a; }
template <int K> DAS_HD Dual<K>& operator-=(Dual<K>& a, const Dual<K>& b) { a = a - b; return a; }
template <int K> DAS_HD Dual<K>& operator+=(Dual<K>& a, double b) { a.v += b; return a; }
template <int K> DAS_HD Dual<K>& operator-=(Dual<K>& a, double b) { a.v -= b; return a; }
template <int K> DAS_HD Dual<K>& operator*=(Dual<K>& a, const Dual<K>& b) { a = a * b; return a; }
template <int K> DAS_HD Dual<K>& operator*=(Dual<K>& a, double b) { a = a * b; return a; }

// value with a prescribed first tangent (forward-mode seeding of parameters, e.g. boundary values)
template <class T> struct MkSeed;
template <> struct MkSeed<double> { static DAS_HD double make(double v, double) { return v; } };
template <int K> struct MkSeed<Dual<K>> {
    static DAS_HD Dual<K> make(double v, double dv) { Dual<K> r(v); r.d[0] = dv; return r; }
    // the value may carry tangents itself (a patch velocity computed from perturbed metrics): the seed is added to them
    static DAS_HD Dual<K> make(const Dual<K>& v, double dv) { Dual<K> r = v; r.d[0] += dv; return r; }
};

// elementary functions (generic names usable with double as well)
DAS_HD double dsqrt(double a) { return sqrt(a); }
DAS_HD double dexp(double a) { return exp(a); }
DAS_HD double dpow(double a, double e) { return pow(a, e); }
DAS_HD double dabs(double a) { return fabs(a); }
DAS_HD double dmax(double a, double b) { return a >= b ? a : b; }
DAS_HD double dmin(double a, double b) { return a <= b ? a : b; }

template <int K>
DAS_HD Dual<K> dsqrt(const Dual<K>& a) {
    Dual<K> r;
    r.v = sqrt(a.v);
    double g = a.v > 0.0 ? 0.5 / r.v : 0.0;
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = g * a.d[k];
    return r;
}
template <int K>
DAS_HD Dual<K> dexp(const Dual<K>& a) {
    Dual<K> r;
    r.v = exp(a.v);
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = r.v * a.d[k];
    return r;
}
template <int K>
DAS_HD Dual<K> dpow(const Dual<K>& a, double e) {
    Dual<K> r;
    r.v = pow(a.v, e);
    double g = e * pow(a.v, e - 1.0);
#pragma unroll
    for (int k = 0; k < K; k++) r.d[k] = g * a.d[k];
    return r;
}
template <int K> DAS_HD Dual<K> dabs(const Dual<K>& a) { return a.v >= 0.0 ? a : -a; }
// max/min select on the value (ties -> first argument, like the oracle's real-part compare)
template <int K> DAS_HD Dual<K> dmax(const Dual<K>& a, const Dual<K>& b) { return a.v >= b.v ? a : b; }
template <int K> DAS_HD Dual<K> dmin(const Dual<K>& a, const Dual<K>& b) { return a.v <= b.v ? a : b; }
template <int K> DAS_HD Dual<K> dmax(const Dual<K>& a, double b) { return a.v >= b ? a : Dual<K>(b); }
template <int K> DAS_HD Dual<K> dmin(const Dual<K>& a, double b) { return a.v <= b ? a : Dual<K>(b); }

}  // namespace das
