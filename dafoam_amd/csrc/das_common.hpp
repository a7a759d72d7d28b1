// Shared declarations of the MI355X-native adjoint hot path (host side).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <omp.h>
#include <sched.h>
#include <mutex>
#include <thread>
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

// ---- host threads ---------------------------------------------------------------------------
// CPUs this process may really use: the affinity mask AND the container's CFS quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us).  The
// round-4 bench host shows 256 CPUs to a container whose quota is 16: OpenMP regions with 256 threads then run throttled (host
// STREAM 33 GB/s instead of 488, profiles/r05m_host_cpu.txt).  das_create caps the OpenMP team size of the host phases (connectivity
// pattern, prune, structure builders) to this number once per process.
inline int usable_host_cpus() {
    int n = omp_get_num_procs();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) n = std::min(n, c); }
    double quota = -1.0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0}; double per = 0.0;
        if (fscanf(f, "%63s %lf", q, &per) == 2 && std::string(q) != "max" && per > 0.0) quota = atof(q) / per;
        fclose(f);
    } else {
        double q = -1.0, per = 0.0;
        if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f1, "%lf", &q) != 1) q = -1.0; fclose(f1); }
        if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lf", &per) != 1) per = 0.0; fclose(f2); }
        if (q > 0.0 && per > 0.0) quota = q / per;
    }
    if (quota > 0.0) n = std::min(n, std::max(1, (int)(quota + 0.5)));
    return std::max(1, n);
}
// The cap is a number the host-phase parallel regions pass as num_threads(...) (host_threads()); the OpenMP runtime's own
// num-threads ICV is per thread and belongs to the embedding application (torch, numpy): the library does not touch it
// (ADVICE round 4).
inline int host_thread_cap() {
    static const int cap = getenv("DAS_KEEP_OMP_THREADS") ? omp_get_num_procs() : usable_host_cpus();
    return cap;
}
inline int host_threads(int want = 1 << 30) { return std::max(1, std::min(std::min(want, host_thread_cap()), omp_get_max_threads())); }

// ---- device buffer -------------------------------------------------------------------------
// mapped virtual ranges of released VmBufs (never unmapped; see VmBuf::release)
struct VmRange {
    void* p = nullptr;
    size_t reservedBytes = 0, mappedBytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
    int device = 0;
};
inline std::vector<VmRange>& vm_cache() { static std::vector<VmRange> c; return c; }
inline std::mutex& vm_cache_mutex() { static std::mutex m; return m; }
inline std::recursive_mutex& vm_api_mutex() { static std::recursive_mutex m; return m; }  // recursive: vm_cache_trim is also called by the mapper, which holds it
// last resort of a failed hipMalloc: give the physical memory of the cached (idle) ranges back to the device.  The ranges are
// dropped for good - a range that was unmapped is never handed out again (see VmBuf::release)
// (only the ranges of `device` - the other GPUs' solvers keep theirs; the VM API calls are serialised with the mappers)
inline size_t vm_cache_trim(int device = -1) {
    std::lock_guard<std::recursive_mutex> api(vm_api_mutex());
    std::lock_guard<std::mutex> lk(vm_cache_mutex());
    if (device < 0 && hipGetDevice(&device) != hipSuccess) device = -1;
    size_t freed = 0;
    std::vector<VmRange> keep;
    for (VmRange& r : vm_cache()) {
        if (device >= 0 && r.device != device) { keep.push_back(std::move(r)); continue; }
        size_t off = 0;
        for (size_t i = 0; i < r.handles.size(); i++) {
            (void)hipMemUnmap((char*)r.p + off, r.sizes[i]);
            (void)hipMemRelease(r.handles[i]);
            off += r.sizes[i];
            freed += r.sizes[i];
        }
        // the address range itself stays reserved (never freed: a later reservation must not land on it)
    }
    vm_cache().swap(keep);
    return freed;
}

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
        if (!n) return;
        if (hipMalloc((void**)&p, n * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            // out of memory: idle Krylov ranges of destroyed solvers may still hold mapped memory (VmBuf cache) - release it and retry
            if (vm_cache_trim() > 0) (void)hipDeviceSynchronize();
            DAS_HIP(hipMalloc((void**)&p, n * sizeof(T)));
        }
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

// ---- large device buffer through the virtual-memory API ----------------------------------------------------------
// The Krylov basis of the reference's default restart (1000 vectors) is 129 GB at 2 M cells.  Measured on the MI355X
// (tools/gpu/alloc_bench.hip, profiles/r03h_alloc_bench.log; bench runs r03g / r03i): getting such a block right after other
// multi-GB buffers were freed (the assembly maps, a smaller basis) stalls for 3-5 s - freed HBM is scrubbed before it is handed
// out again - whether it comes from hipMalloc or from hipMemCreate.  So the basis is never allocated in one piece and never
// re-allocated: the address range for the largest basis the memory budget allows is RESERVED once (free), and 2 GB physical
// chunks are mapped on demand while the iteration advances (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess):
// a solve that converges after 464 vectors maps 60 GB, not 129, the mapping of the next chunk costs milliseconds, and the
// scrubbing of freed memory overlaps with the first iterations.  The kernels see one contiguous range.  Any failure of the VM
// path falls back to one hipMalloc of the whole range.

template <class T>
struct VmBuf {
    T* p = nullptr;
    size_t n = 0;        // reserved elements (the capacity the kernels may address once mapped)
    size_t mappedBytes = 0;
    bool vmm = false;
    size_t reservedBytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
    hipMemAllocationProp prop = {};
    static constexpr size_t CHUNK = (size_t)2 << 30;
    size_t granularity = (size_t)2 << 20;
    // chunks are mapped by a helper thread AHEAD of the iteration (request), the solver only waits if it catches up (ensure):
    // hipMemCreate can block for tens of milliseconds while freed HBM is still being scrubbed
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    size_t targetBytes = 0;
    bool stopping = false;
    int device = 0;
    std::string workerError;
    VmBuf() = default;
    VmBuf(const VmBuf&) = delete;
    VmBuf& operator=(const VmBuf&) = delete;
    ~VmBuf() { release(); }
    void stop_worker() {
        if (worker.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); stopping = true; }
            cv.notify_all();
            worker.join();
        }
        stopping = false;
    }
    void release() {
        stop_worker();
        if (vmm && p) {
            // the range is NOT unmapped: it goes to a process-wide cache and is handed to the next buffer that fits (round 4: a
            // range reserved AFTER an earlier one had been unmapped and freed faulted as soon as its second chunk was written -
            // "write access to a read-only page", twice, at 200 k cells - and re-mapping gigabytes per solver was pure overhead)
            int cur = -1;
            (void)hipGetDevice(&cur);
            if (cur != device) (void)hipSetDevice(device);  // (the range lives on `device`, not on the caller's current one)
            (void)hipDeviceSynchronize();  // nothing in flight may still address the range when its next owner starts writing
            if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
            VmRange r;
            r.p = (void*)p; r.reservedBytes = reservedBytes; r.mappedBytes = mappedBytes; r.handles = handles; r.sizes = sizes; r.device = device;
            std::lock_guard<std::mutex> lk(vm_cache_mutex());
            vm_cache().push_back(std::move(r));
        } else if (p) {
            (void)hipFree(p);
        }
        handles.clear(); sizes.clear();
        p = nullptr; n = 0; vmm = false; reservedBytes = 0; mappedBytes = 0; targetBytes = 0; workerError.clear();
    }
    // reserve the address range for n_ elements (nothing is mapped yet on the VM path)
    void reserve(size_t n_) {
        release();
        if (!n_) return;
        const size_t bytes = n_ * sizeof(T);
        if (bytes >= ((size_t)4 << 30) && !getenv("DAS_NO_VMM") && hipGetDevice(&device) == hipSuccess) {
            prop = hipMemAllocationProp{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = device;
            size_t gran = 0;
            if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0) {
                const size_t align = std::max<size_t>(gran, (size_t)2 << 20);
                granularity = align;
                // whole 2 GB chunks only (address space is free; mapping a partial last chunk failed in hipMemSetAccess on ROCm 7.2)
                const size_t unit = std::max(align, CHUNK / align * align);
                const size_t total = (bytes + unit - 1) / unit * unit;
                {   // a cached range of this device that is large enough (smallest fit), with whatever it has mapped already
                    std::lock_guard<std::mutex> lk(vm_cache_mutex());
                    std::vector<VmRange>& cache = vm_cache();
                    int best = -1;
                    for (size_t i = 0; i < cache.size(); i++)
                        if (cache[i].device == device && cache[i].reservedBytes >= total && (best < 0 || cache[i].reservedBytes < cache[best].reservedBytes)) best = (int)i;
                    if (best >= 0) {
                        VmRange r = std::move(cache[best]);
                        cache.erase(cache.begin() + best);
                        p = (T*)r.p; reservedBytes = r.reservedBytes; mappedBytes = r.mappedBytes; handles = std::move(r.handles); sizes = std::move(r.sizes);
                        vmm = true; n = n_;
                        worker = std::thread([this]() { this->map_loop(); });
                        return;
                    }
                }
                // nothing cached fits: the idle ranges hold physical memory this (larger) basis will need - give it back to the device
                // (a later, larger solver otherwise runs out of memory next to tens of idle GB)
                if (vm_cache_trim(device) > 0) (void)hipDeviceSynchronize();
                void* base = nullptr;
                if (hipMemAddressReserve(&base, total, align, nullptr, 0) == hipSuccess && base) {
                    p = (T*)base; reservedBytes = total; vmm = true; n = n_;
                    worker = std::thread([this]() { this->map_loop(); });
                    return;
                }
                (void)hipGetLastError();
            }
        }
        DAS_HIP(hipMalloc((void**)&p, bytes));  // small buffers (fast) or no VM support
        n = n_; mappedBytes = bytes;
    }
    void map_loop() {
        (void)hipSetDevice(device);
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&]() { return stopping || (mappedBytes < targetBytes && workerError.empty()); });
            if (stopping) return;
            const size_t off = mappedBytes;
            size_t sz = std::min(CHUNK, reservedBytes - off);
            lk.unlock();
            // (one mapper at a time, process-wide: concurrent hipMemCreate / hipMemMap / hipMemSetAccess from two threads were the one thing the
            //  two failing split-basis runs of round 5 had that the fp64 runs never had)
            std::lock_guard<std::recursive_mutex> vmLock(vm_api_mutex());
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, sz, &prop, 0);
            // (a 2 GB physical chunk may not exist in fragmented HBM although smaller ones do)
            while (e != hipSuccess && sz > ((size_t)128 << 20) && sz % granularity == 0) {
                (void)hipGetLastError();
                sz = std::max((size_t)128 << 20, sz / 4 / granularity * granularity);
                e = hipMemCreate(&h, sz, &prop, 0);
            }
            if (e != hipSuccess) {
                // idle cached ranges of destroyed solvers may hold the memory this chunk needs (the range in use is not in the cache)
                (void)hipGetLastError();
                if (vm_cache_trim(device) > 0) {
                    (void)hipDeviceSynchronize();
                    sz = std::min(CHUNK, reservedBytes - off);
                    e = hipMemCreate(&h, sz, &prop, 0);
                }
            }
            const char* what = "hipMemCreate";
            bool created = e == hipSuccess;
            if (created) { e = hipMemMap((char*)p + off, sz, 0, h, 0); what = "hipMemMap"; }
            if (e == hipSuccess) { e = hipMemSetAccess((char*)p + off, sz, &acc, 1); what = "hipMemSetAccess"; }
            lk.lock();
            if (e != hipSuccess) {
                if (created) (void)hipMemRelease(h);
                (void)hipGetLastError();
                size_t fr = 0, tot = 0;
                (void)hipMemGetInfo(&fr, &tot);
                workerError = std::string("mapping more of the Krylov basis failed (") + what + ": " + hipGetErrorString(e) + "; " +
                              std::to_string(mappedBytes >> 20) + " MiB mapped, chunk " + std::to_string(sz >> 20) + " MiB, device memory free " +
                              std::to_string(fr >> 20) + " of " + std::to_string(tot >> 20) + " MiB)";
            } else {
                handles.push_back(h); sizes.push_back(sz);
                mappedBytes = off + sz;
            }
            cv.notify_all();
        }
    }
    // ask for the first `count` elements to become addressable (asynchronous)
    void request(size_t count) {
        if (!vmm) return;
        const size_t need = std::min(reservedBytes, count * sizeof(T));
        { std::lock_guard<std::mutex> lk(mu); if (need > targetBytes) targetBytes = need; }
        cv.notify_all();
    }
    // wait until the first `count` elements are addressable; false: the device has no memory left for them (what is mapped stays
    // usable - the Krylov solver closes its cycle there)
    bool try_ensure(size_t count) {
        if (!vmm) return count <= n;
        request(count);
        const size_t need = std::min(reservedBytes, count * sizeof(T));
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return mappedBytes >= need || !workerError.empty(); });
        return mappedBytes >= need;
    }
    void ensure(size_t count) {
        if (!try_ensure(count)) throw Error(DAS_ERR_INTERNAL, workerError.empty() ? std::string("Krylov basis: request beyond the reserved range") : workerError);
    }
    void alloc(size_t n_) { reserve(n_); ensure(n_); }
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
// (templated on the scalar: double everywhere except in the mesh-sensitivity pass, where the metrics carry tangents - Dual<1>)
template <class G>
struct FaceGeomT {     // 12 scalars (double: 96 B per face)
    G Sf[3];           // area vector owner -> neighbour (outward on boundary)
    G magSf;
    G w;               // linear interpolation weight of the owner value (1 on boundary)
    G nod;             // nonOrthDeltaCoeffs (boundary: deltaCoeffs = 1/|Cf - C|)
    G corr[3];         // nonOrthCorrectionVectors (0 on boundary)
    G Cf[3];
};
template <class G>
struct CellGeomT {  // 5 scalars
    G C[3];
    G V;
    G y;  // frozen wall distance
};
typedef FaceGeomT<double> FaceGeom;
typedef CellGeomT<double> CellGeom;

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
    struct GeomTopo geom_topo() const;  // das_geom.hpp
    void compute_geometry(const double* y_wall);
    void build_addressing();
};

// mesh points: the cells whose residual rows feel a point, a colouring of the points with pairwise disjoint sets, FD steps
struct PointInfluence {
    int rings = 0, nColors = 0;
    std::vector<long long> ptr;      // nP + 1
    std::vector<int> cells;          // influenced cells of every point, ascending
    std::vector<int> color;          // nP (-1: a point no face uses)
    std::vector<int> cptr, cpoints;  // colour -> points
    std::vector<double> h;           // nP: central-difference step of the point (point_steps)
};
void build_point_influence(const Mesh& m, int rings, int threads, PointInfluence& out);
void point_steps(const Mesh& m, double relStep, std::vector<double>& h);

// aggregates along the strongest pressure-Laplacian couplings (at most maxAgg of them); ownedCell: optional mask; returns the count
int strength_aggregates(const Mesh& m, const std::vector<unsigned char>* ownedCell, int maxAgg, std::vector<int>& agg);

void mesh_metrics_only(int nP, const double* pts, int nF, int nIF, int nC, const int* fptr, const int* fpts, const int* own, const int* nei,
                       double* Sf, double* Cf, double* C, double* V, double* w);
double wall_seconds();

}  // namespace das
