// Shared declarations of the MI355X-native adjoint hot path (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dafoam_amd.h"

namespace das {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define DAS_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw das::Error(DAS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" \
                                              + __FILE__ + ":" + std::to_string(__LINE__));        \
    } while (0)

#define DAS_CHECK(cond, code, msg)                    \
    do {                                              \
        if (!(cond)) throw das::Error((code), (msg)); \
    } while (0)

// ---- device buffer -------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t n_) { alloc(n_); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t n_) {
        release();
        n = n_;
        if (n) DAS_HIP(hipMalloc((void**)&p, n * sizeof(T)));
    }
    void upload(const T* h, size_t cnt) {
        if (cnt > n) alloc(cnt);
        if (cnt) DAS_HIP(hipMemcpy(p, h, cnt * sizeof(T), hipMemcpyHostToDevice));
    }
    void upload(const std::vector<T>& h) { upload(h.data(), h.size()); }
    void download(T* h, size_t cnt) const {
        if (cnt) DAS_HIP(hipMemcpy(h, p, cnt * sizeof(T), hipMemcpyDeviceToHost));
    }
    std::vector<T> to_host() const {
        std::vector<T> h(n);
        download(h.data(), n);
        return h;
    }
    void zero() {
        if (n) DAS_HIP(hipMemset(p, 0, n * sizeof(T)));
    }
};

// ---- options: flattened DAOPTION keys (reference dafoam/pyDAFoam.py:39-661) ---------------------
struct Options {
    std::map<std::string, double> d;
    std::map<std::string, long long> i;
    std::map<std::string, std::string> s;
    Options();
    double getd(const std::string& k) const;
    long long geti(const std::string& k) const;
    const std::string& gets(const std::string& k) const;
    bool list_has(const std::string& k, const std::string& item) const;
};

// ---- face / cell records as consumed by the kernels (AoS records, fully consumed per access) ----
struct FaceGeom {      // 12 doubles = 96 B per face
    double Sf[3];      // area vector owner -> neighbour (outward on boundary)
    double magSf;
    double w;          // linear interpolation weight of the owner value (1 on boundary)
    double nod;        // nonOrthDeltaCoeffs (boundary: deltaCoeffs = 1/|Cf - C|)
    double corr[3];    // nonOrthCorrectionVectors (0 on boundary)
    double Cf[3];
};
struct CellGeom {  // 5 doubles
    double C[3];
    double V;
    double y;  // frozen wall distance
};

struct PatchBC {  // per patch, small table
    int type;
    int U_code, p_code, nuTilda_code, nut_code, T_code;
    double U_val[3];
    double p_val, nuTilda_val, T_val;
    // tangent of the patch values (forward-mode seed for dR/d(BC value); zero except inside das_calc_drdbc)
    double dU_val[3];
    double dp_val, dnuTilda_val, dT_val;
    int mrf_included;  // 1 = the patch rotates with the MRF zone (MRFZone includedFaces)
    int rot;           // cyclic patch with a rotation: neighbour-side vectors are multiplied by Q (forwardT)
    double Q[9];
};

// ---- host mesh (fvMesh equivalent) ------------------------------------------------------------
struct Mesh {
    int nP = 0, nF = 0, nIF = 0, nC = 0, nPatch = 0;
    std::vector<double> points;
    std::vector<int> face_ptr, face_pts, owner, neighbour;
    std::vector<int> patch_start, patch_size, patch_type;
    std::vector<PatchBC> bc;
    // geometry
    std::vector<FaceGeom> fg;
    std::vector<CellGeom> cg;
    std::vector<int> bface_patch;  // nBF
    std::vector<int> cyc_face;     // nBF: the paired face of a cyclic boundary face, -1 otherwise
    // cell -> faces CSR; entry = face id | (side<<31), side 1 = this cell is the face's neighbour
    std::vector<int> cf_ptr, cf_face, cf_other;
    // cell -> cells CSR (face neighbours, ascending)
    std::vector<int> cc_ptr, cc;
    void build(const das_case_t* c);
    void compute_geometry(const double* y_wall);
    void build_addressing();
};

double wall_seconds();

}  // namespace das
