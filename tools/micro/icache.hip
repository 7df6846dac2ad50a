// How much does a cold instruction cache add to a kernel?  K<KB,ID>: KB kilobytes of straight-line VALU code run once per wave.
// "warm": the same kernel launched repeatedly; "cold": four distinct kernels of the same size in rotation.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KB, int ID>
__global__ __launch_bounds__(256) void K(float* out, float s) {
    float x = s + threadIdx.x;
#pragma unroll
    for (int i = 0; i < KB * 128; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(s));
    if (x == 12345.f) out[ID] = x;
}
template <int KB>
void run(float* d, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int L = 200;
    float warm, cold;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, st);
        for (int i = 0; i < L; ++i) K<KB, 0><<<256, 256, 0, st>>>(d, 0.5f);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&warm, e0, e1);
        hipEventRecord(e0, st);
        for (int i = 0; i < L / 4; ++i) {
            K<KB, 0><<<256, 256, 0, st>>>(d, 0.5f);
            K<KB, 1><<<256, 256, 0, st>>>(d, 0.5f);
            K<KB, 2><<<256, 256, 0, st>>>(d, 0.5f);
            K<KB, 3><<<256, 256, 0, st>>>(d, 0.5f);
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&cold, e0, e1);
    }
    printf("%3d KB: same kernel %.2f us/launch, rotating 4 kernels %.2f us/launch\n", KB, warm * 1e3 / L, cold * 1e3 / L);
}
int main() {
    float* d; hipMalloc(&d, 1024);
    hipStream_t st; hipStreamCreate(&st);
    run<1>(d, st); run<4>(d, st); run<8>(d, st); run<16>(d, st); run<24>(d, st); run<32>(d, st); run<48>(d, st);
    return 0;
}
