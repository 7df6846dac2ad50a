// Depthwise 3x3 convolution (stride 1, padding 1) on NHWC bf16: an HBM-bound 9-tap stencil, not a GEMM.
// One thread produces 8 consecutive channels of one pixel from nine 16-byte neighbour loads (the 3x reuse
// along each axis is served by L1/L2; the tensor is read from HBM once).  The data-gradient is the same
// kernel with tap-flipped weights.  The weight-gradient is a per-channel 9-tap reduction over all pixels:
// each thread keeps 72 + 8 partial sums for its 8 channels over a strip of pixels, lanes that share a
// channel vector are combined with wave shuffles, and one atomic per (tap, channel) per wave lands in HBM.
#include "common.h"
#include "ddpm_hip.h"

__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, bf16* __restrict__ y, int N, int H, int W,
                                                         int C) {
    const int cvs = C / 8;
    const long long total = (long long)N * H * W * cvs;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % cvs);
        const long long pix = i / cvs;
        const int wx = (int)(pix % W);
        const int hy = (int)((pix / W) % H);
        const int c = cv * 8;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias ? bias[c + j] : 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = hy + kh - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = wx + kw - 1;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *(const bf16x8*)(x + (pix + (kh - 1) * W + (kw - 1)) * C + c);
                const float* wp = w + (kh * 3 + kw) * C + c;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j] * wp[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        *(bf16x8*)(y + pix * C + c) = o;
    }
}

extern "C" int ddpm_dwconv3x3_forward(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                                      ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && w && y, "dwconv: null pointer");
    DDPM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "dwconv: C (%d) must be a positive multiple of 8", C);
    const long long total = (long long)N * H * W * (C / 8);
    long long grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C);
    return ddpm_check_launch("dwconv3x3");
}

// dw[tap][c] += sum_{n,h,w} dy[n,h,w,c] * x[n,h+kh-1,w+kw-1,c];  db[c] += sum dy.
// Thread layout: 256 threads = (256/cvs) pixel lanes x cvs channel vectors (cvs = C/8, a power of two <= 64).
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                               float* __restrict__ dw, float* __restrict__ db, int N, int H, int W,
                                                               int C, int pix_per_block) {
    const int cvs = C / 8;
    const int cv = threadIdx.x % cvs, prow = threadIdx.x / cvs, pstep = 256 / cvs;
    const long long npix = (long long)N * H * W;
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > npix) p1 = npix;
    const int c = cv * 8;
    float acc[9][8], sb[8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) sb[j] = 0.f;
    for (long long pix = p0 + prow; pix < p1; pix += pstep) {
        const int wx = (int)(pix % W);
        const int hy = (int)((pix / W) % H);
        const bf16x8 g = *(const bf16x8*)(dy + pix * C + c);
        float gf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gf[j] = (float)g[j]; sb[j] += gf[j]; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = hy + kh - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = wx + kw - 1;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *(const bf16x8*)(x + (pix + (kh - 1) * W + (kw - 1)) * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[kh * 3 + kw][j] += gf[j] * (float)v[j];
            }
        }
    }
    // combine the lanes of a wave that hold the same channel vector, then one atomic per wave
    const int lane = threadIdx.x & 63;
    const int lo = cvs < 64 ? cvs : 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[t][j];
            for (int o = 32; o >= lo; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < lo) atomicAdd(dw + t * C + c + j, v);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = sb[j];
        for (int o = 32; o >= lo; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane < lo && db) atomicAdd(db + c + j, v);
    }
}

extern "C" int ddpm_dwconv3x3_wgrad(const void* x, const void* dy, float* dw, float* db, int N, int H, int W, int C,
                                    ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && dy && dw, "dwconv_wgrad: null pointer");
    const int cvs = C / 8;
    DDPM_REQUIRE(N > 0 && H > 0 && W > 0 && C % 8 == 0 && cvs >= 1 && cvs <= 64 && (cvs & (cvs - 1)) == 0,
                 "dwconv_wgrad: C (%d) must be 8 * a power of two <= 512", C);
    const long long npix = (long long)N * H * W;
    // enough blocks to fill the chip, each with a strip long enough to amortise its 80 atomics per wave lane-group
    long long blocks = (npix * cvs + 256 * 32 - 1) / (256 * 32);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    const int ppb = (int)((npix + blocks - 1) / blocks);
    blocks = (npix + ppb - 1) / ppb;
    hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)x, (const bf16*)dy, dw, db, N, H,
                       W, C, ppb);
    return ddpm_check_launch("dwconv3x3_wgrad");
}
