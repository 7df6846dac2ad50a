// Machine probe: shader clock under load, dependent-load latency, streaming copy bandwidth, trivial-kernel duration.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void clock_probe(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}
__global__ void chase(const int* next, int* out, int steps, unsigned long long* cyc) {
    int p = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; ++i) p = next[p];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    *out = p; *cyc = t1 - t0;
}
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void trivial(int* p) { if (threadIdx.x == 0) *p += 1; }

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    unsigned long long h[4];
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(clock_probe, dim3(1024), dim3(256), 0, 0, d, 2000000);
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("clock: %.0f MHz (memtime %llu / memrealtime %llu @100MHz)\n", (double)h[0] / h[1] * 100.0, h[0], h[1]);
    }
    // pointer chase over 64 MB (beyond L2, inside Infinity Cache) and 1 GB (HBM), stride 4 KB-ish random
    for (size_t mb : {1, 64, 1024}) {
        size_t n = mb * 1024 * 1024 / 4;
        std::vector<int> nx(n, 0);
        size_t stride = 1031 * 16;   // ints; ~66 KB
        size_t p = 0;
        for (size_t i = 0; i < 200000; ++i) { size_t q = (p + stride) % n; nx[p] = (int)q; p = q; }
        int* dn; int* dout; CK(hipMalloc(&dn, n * 4)); CK(hipMalloc(&dout, 4));
        CK(hipMemcpy(dn, nx.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, dn, dout, 20000, d);
        hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, dn, dout, 20000, d);
        CK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
        printf("dependent-load latency over %zu MB: %.0f cycles\n", mb, (double)h[0] / 20000);
        hipFree(dn); hipFree(dout);
    }
    // streaming copy 2 x 1 GB
    long long n4 = 1ll << 26;  // 1 GiB of float4
    float4 *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 1, n4 * 16));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(copy4, dim3(256 * 16), dim3(256), 0, 0, a, b, n4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy 1 GiB -> 1 GiB: %.3f ms = %.2f TB/s (read+write)\n", ms, 2.0 * n4 * 16 / ms / 1e9);
    }
    // small copy: 8 MB (L2/MALL resident) repeated
    for (long long bytes : {1ll << 20, 8ll << 20, 64ll << 20}) {
        long long m4 = bytes / 16;
        hipLaunchKernelGGL(copy4, dim3(1024), dim3(256), 0, 0, a, b, m4);
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(copy4, dim3(1024), dim3(256), 0, 0, a, b, m4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy %lld MB x50 back-to-back: %.2f us each = %.2f TB/s\n", bytes >> 20, ms * 1000 / 50, 2.0 * bytes * 50 / ms / 1e9);
    }
    int* cnt; CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4));
    hipEventRecord(e0);
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(trivial, dim3(1), dim3(64), 0, 0, cnt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("1000 dependent trivial kernels: %.2f us each\n", ms);
    return 0;
}
