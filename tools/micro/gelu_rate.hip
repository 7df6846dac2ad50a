// VALU cost of the activation arithmetic, no memory traffic: every thread evaluates ITER x 8 values held in registers.
// Prints nanoseconds per launch and the implied SIMD cycles per wave-wide evaluation (at the clock given as argv[1], GHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I diffusion-model_amd/csrc -I include tools/micro/gelu_rate.hip -o tools/micro/gelu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "common.h"

void ddpm_set_error(const char*, ...) {}
int ddpm_check_launch(const char*) { return 0; }

constexpr int ITER = 512;

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float seed) {
    float v[8], acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = seed + 0.001f * (threadIdx.x + j); acc[j] = 0.f; }
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float r;
            if (MODE == 0) r = gelu_fwd(v[j]);
            else if (MODE == 1) r = act_bwd(v[j], ACT_GELU);
            else if (MODE == 2) r = __builtin_amdgcn_exp2f(v[j]);
            else if (MODE == 3) r = __builtin_amdgcn_rcpf(v[j]);
            else if (MODE == 4) r = v[j] * 1.0001f + 0.5f;
            else r = act_fwd(v[j], ACT_SILU);
            acc[j] += r;
            v[j] = v[j] * 0.999f + 1e-3f * r;      // the next value depends on this one: nothing is hoisted
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, double ghz) {
    float* out;
    hipMalloc(&out, 4096);
    const int blocks = 256 * 8;                     // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    rate_kernel<MODE><<<blocks, 256>>>(out, 0.3f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) rate_kernel<MODE><<<blocks, 256>>>(out, 0.3f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / 5;
    // wave-wide evaluations per SIMD: blocks * 4 waves * ITER * 8 / (256 CUs * 4 SIMDs)
    const double evals = (double)blocks * 4 * ITER * 8 / 1024.0;
    printf("%-10s %9.1f us   %6.1f cycles per wave-wide evaluation (incl. 3 bookkeeping ops)\n", name, ns / 1e3, ns * ghz / evals);
    hipFree(out);
}

int main(int argc, char** argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    run<4>("fma", ghz);
    run<2>("exp2", ghz);
    run<3>("rcp", ghz);
    run<0>("gelu", ghz);
    run<1>("gelu'", ghz);
    run<5>("silu", ghz);
    return 0;
}
