// How long does it take to get N GB of HBM?  hipMalloc vs the virtual-memory API (reserve once, map physical chunks on demand).
// hipcc --offload-arch=gfx950 -O2 -o alloc_bench tools/gpu/alloc_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t GB = 1ull << 30;
    for (size_t gb : {8ull, 32ull, 128ull}) {
        void* p = nullptr;
        double t = now();
        CK(hipMalloc(&p, gb * GB));
        double t1 = now();
        CK(hipMemset(p, 0, gb * GB));
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipMemset(p, 1, gb * GB));
        CK(hipDeviceSynchronize());
        double t3 = now();
        CK(hipFree(p));
        double t4 = now();
        printf("hipMalloc %3zu GB: malloc %.3f s, first memset %.3f s, second memset %.3f s, free %.3f s\n", gb, t1 - t, t2 - t1, t3 - t2, t4 - t3);
    }
    int dev = 0;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("vmm granularity %zu\n", gran);
    const size_t total = 128 * GB, chunk = 4 * GB;
    void* base = nullptr;
    double t = now();
    CK(hipMemAddressReserve(&base, total, gran, nullptr, 0));
    printf("reserve 128 GB: %.4f s\n", now() - t);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    double tc = 0, tm = 0, ta = 0;
    for (size_t off = 0; off < 64 * GB; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        double a = now();
        CK(hipMemCreate(&h, chunk, &prop, 0));
        double b = now();
        CK(hipMemMap((char*)base + off, chunk, 0, h, 0));
        double c = now();
        CK(hipMemSetAccess((char*)base + off, chunk, &acc, 1));
        double d = now();
        tc += b - a; tm += c - b; ta += d - c;
        hs.push_back(h);
    }
    printf("vmm 16 x 4 GB: create %.3f s, map %.3f s, setaccess %.3f s\n", tc, tm, ta);
    t = now();
    CK(hipMemset(base, 0, 64 * GB));
    CK(hipDeviceSynchronize());
    printf("memset 64 GB through the mapping: %.3f s\n", now() - t);
    for (size_t i = 0; i < hs.size(); i++) { CK(hipMemUnmap((char*)base + i * chunk, chunk)); CK(hipMemRelease(hs[i])); }
    CK(hipMemAddressFree(base, total));
    return 0;
}
