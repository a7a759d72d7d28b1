// PMC calibration: what does FETCH_SIZE report for a PURE STREAM whose bytes are known, as a function of the
// access width per lane?  Three read-only streaming kernels (4, 8, 16 bytes per lane and load, grid-stride, fully coalesced) and one
// write kernel over N bytes each; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and compare Counter_Value x 1024 with N
// (tools/pmc_summary.py --calib).  hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void calib_read_4B(const float* __restrict__ a, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456f) out[0] = s;
}
__global__ void calib_read_8B(const double* __restrict__ a, size_t n, double* out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456) out[0] = s;
}
__global__ void calib_read_16B(const double2* __restrict__ a, size_t n, double* out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
// gather of 8-byte values through an index stream (the shape of the SpMV's x gather): idx is a random permutation inside windows of W
__global__ void calib_gather_8B(const double* __restrict__ x, const int* __restrict__ idx, size_t n, double* out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[idx[i]];
    if (s == 123.456) out[0] = s;
}
__global__ void calib_write_8B(double* __restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = 1.0;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 8) << 30;  // GiB per stream
    void* buf; double* out; int* idx;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    const size_t ng = (size_t)1 << 28;  // 2^28 gathered values (1 GiB of indices, 2 GiB of x)
    CK(hipMalloc(&idx, ng * sizeof(int)));
    {
        int* h = (int*)malloc(ng * sizeof(int));
        unsigned long long st = 88172645463325252ull;
        const size_t W = 4096;  // permute inside windows of 4096 entries (32 KB of x): every x value is used exactly once
        for (size_t b = 0; b < ng; b += W) {
            for (size_t k = 0; k < W; k++) h[b + k] = (int)(b + k);
            for (size_t k = W - 1; k > 0; k--) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; size_t j = st % (k + 1); int t = h[b + k]; h[b + k] = h[b + j]; h[b + j] = t; }
        }
        CK(hipMemcpy(idx, h, ng * sizeof(int), hipMemcpyHostToDevice));
        free(h);
    }
    const int grid = 256 * 16, block = 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_read_4B, dim3(grid), dim3(block), 0, 0, (const float*)buf, bytes / 4, (float*)out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("calib_read_4B   %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_read_8B, dim3(grid), dim3(block), 0, 0, (const double*)buf, bytes / 8, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("calib_read_8B   %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_read_16B, dim3(grid), dim3(block), 0, 0, (const double2*)buf, bytes / 16, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("calib_read_16B  %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_gather_8B, dim3(grid), dim3(block), 0, 0, (const double*)buf, idx, ng, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("calib_gather_8B %zu bytes (4 B index + 8 B value per entry)  %.3f ms  %.0f GB/s\n", ng * 12, ms, ng * 12 / ms / 1e6);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(calib_write_8B, dim3(grid), dim3(block), 0, 0, (double*)buf, bytes / 8); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("calib_write_8B  %zu bytes  %.3f ms  %.0f GB/s\n", bytes, ms, bytes / ms / 1e6);
    }
    return 0;
}
