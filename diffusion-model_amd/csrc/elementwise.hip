// Fused elementwise kernels of the diffusion process: forward noising (q), ancestral reverse step,
// ELBO-weighted loss forward/backward, time embedding, and the NCHW<->NHWC layout/precision bridges
// at the UNet boundary.  All of them are HBM-streaming kernels: 16-byte vector accesses, grid capped
// at a few blocks per CU with a grid-stride loop.
#include <stdlib.h>
#include "common.h"
#include "ddpm_hip.h"

#include <stdarg.h>
#include <stdio.h>

// q / reverse step are checked bit-for-bit against the reference arithmetic (separately rounded
// multiplies and adds), so the compiler must not contract a*b+c into an FMA in this file.
#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// error plumbing shared by the whole library
static thread_local char g_err[512] = "";

void ddpm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ddpm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ddpm_set_error("%s: %s", what, hipGetErrorString(e));
        return DDPM_EHIP;
    }
    return DDPM_OK;
}

extern "C" const char* ddpm_last_error(void) { return g_err; }
extern "C" int ddpm_abi_version(void) { return 4; }

// Reproducible sums (no counterpart in the reference, which leaves torch's own choice of reduction order alone): when on,
// every launcher whose kernel combines per-workgroup partial sums with atomics runs that kernel as ONE workgroup, so the sum is
// formed in a fixed order (slower; meant for parity runs).  Process-wide; initialised from DDPM_DETERMINISTIC=1.
static int g_deterministic = -1;
extern "C" int ddpm_set_deterministic(int on) {
    const int prev = g_deterministic > 0;
    g_deterministic = on ? 1 : 0;
    return prev;
}
extern "C" int ddpm_deterministic(void) {
    if (g_deterministic < 0) {
        const char* e = getenv("DDPM_DETERMINISTIC");
        g_deterministic = (e && e[0] == '1') ? 1 : 0;
    }
    return g_deterministic;
}

static inline int stream_grid(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

// Per-timestep scalars, one row of 8 floats per t (row 0 is unused: t runs 1..T).
enum { SC_SQRT_AB = 0, SC_SQRT_1MAB = 1, SC_A = 2, SC_B = 3, SC_C = 4, SC_W = 5, SC_STRIDE = 8 };

// ------------------------------------------------------------------------------------------------
// q: x_t = sqrt(abar_t) * x0 + sqrt(1 - abar_t) * eps       (reference ddpm.py:102-106)
// Reads x0 (NCHW fp32) and eps (external tensor, or Philox N(0,1) generated in-kernel), writes x_t as
// NCHW fp32 and/or as the channel-padded NHWC bf16 tensor the stem convolution consumes.
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const long long* __restrict__ t,
                                                        const float* __restrict__ noise, const float* __restrict__ sched,
                                                        float* __restrict__ xt, bf16* __restrict__ xt_nhwc, int cpad, int N, int C,
                                                        int HW, const unsigned long long* __restrict__ seed, unsigned stream_id) {
    const long long chw = (long long)C * HW;
    const long long total4 = (long long)N * chw / 4;
    const unsigned long long sd = seed ? *seed : 0ull;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total4; q += (long long)gridDim.x * blockDim.x) {
        const long long e = q * 4;
        const int n = (int)(e / chw);
        const int ti = (int)t[n];
        const float a = sched[ti * SC_STRIDE + SC_SQRT_AB], b = sched[ti * SC_STRIDE + SC_SQRT_1MAB];
        const f32x4 x = *(const f32x4*)(x0 + e);
        const f32x4 z = noise ? *(const f32x4*)(noise + e) : philox_normal4(sd, stream_id, (unsigned long long)q);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a * x[j] + b * z[j];   // two roundings + one: no FMA (contract off)
        if (xt) *(f32x4*)(xt + e) = o;
        if (xt_nhwc) {
            const long long r = e - (long long)n * chw;
            const int c = (int)(r / HW);
            const int p = (int)(r - (long long)c * HW);  // HW % 4 == 0, so the four elements share n and c
#pragma unroll
            for (int j = 0; j < 4; ++j) xt_nhwc[((long long)n * HW + p + j) * cpad + c] = (bf16)o[j];
        }
    }
}

extern "C" int ddpm_q_sample(const float* x0, const int64_t* t_, const float* noise, const float* sched, float* xt, void* xt_nhwc,
                             int cpad, int N, int C, int HW, const uint64_t* seed_, uint32_t stream_id, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long* t = (const long long*)t_;
    const unsigned long long* seed = (const unsigned long long*)seed_;
    DDPM_REQUIRE(x0 && t && sched && (xt || xt_nhwc), "q_sample: null pointer");
    DDPM_REQUIRE(noise || seed, "q_sample: need an external noise tensor or a device seed");
    DDPM_REQUIRE(N > 0 && C > 0 && HW > 0 && HW % 4 == 0, "q_sample: H*W must be a positive multiple of 4 (got %d)", HW);
    DDPM_REQUIRE(!xt_nhwc || cpad >= C, "q_sample: cpad %d < C %d", cpad, C);
    if (xt_nhwc && cpad > C) {
        if (hipMemsetAsync(xt_nhwc, 0, (size_t)N * HW * cpad * sizeof(bf16), stream) != hipSuccess) return ddpm_check_launch("q_sample memset");
    }
    const long long total4 = (long long)N * C * HW / 4;
    hipLaunchKernelGGL(q_sample_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, stream, x0, t, noise, sched, xt, (bf16*)xt_nhwc,
                       cpad, N, C, HW, seed, stream_id);
    return ddpm_check_launch("q_sample");
}

// ------------------------------------------------------------------------------------------------
// reverse step: x_{t-1} = A_t x_t + B_t x0_pred + C_t z        (reference ddpm.py:138-144)
// A_t = (1-abar_{t-1}) sqrt(alpha_t)/(1-abar_t), B_t = sqrt(abar_{t-1}) beta_t/(1-abar_t),
// C_t = sqrt(1-abar_{t-1}) sqrt(beta_t/(1-abar_t)); C_1 = 0 so the last step adds no noise.
// (out may be xt: every element is read and written by the same thread, and neither pointer is declared restrict)
__global__ __launch_bounds__(256) void reverse_step_kernel(const float* xt, const float* __restrict__ x0p,
                                                            const long long* __restrict__ t, const float* __restrict__ noise,
                                                            const float* __restrict__ sched, float* out, int N, long long chw,
                                                            const unsigned long long* __restrict__ seed, unsigned stream_id,
                                                            int sigmoid_pred, unsigned long long q0) {
    const long long total4 = (long long)N * chw / 4;
    const unsigned long long sd = seed ? *seed : 0ull;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total4; q += (long long)gridDim.x * blockDim.x) {
        const long long e = q * 4;
        const int n = (int)(e / chw);
        const int ti = (int)t[n];
        const float A = sched[ti * SC_STRIDE + SC_A], B = sched[ti * SC_STRIDE + SC_B], Cc = sched[ti * SC_STRIDE + SC_C];
        const f32x4 x = *(const f32x4*)(xt + e);
        f32x4 p = *(const f32x4*)(x0p + e);
        const f32x4 z = noise ? *(const f32x4*)(noise + e) : philox_normal4(sd, stream_id, q0 + (unsigned long long)q);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (sigmoid_pred) p[j] = 1.0f / (1.0f + __expf(-p[j]));
            o[j] = (A * x[j] + B * p[j]) + Cc * z[j];   // same association as the reference, no FMA
        }
        *(f32x4*)(out + e) = o;
    }
}

extern "C" int ddpm_reverse_step(const float* xt, const float* x0_pred, const int64_t* t_, const float* noise, const float* sched,
                                 float* out, int N, int64_t chw, const uint64_t* seed_, uint32_t stream_id, int sigmoid_pred,
                                 uint64_t first_index, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long* t = (const long long*)t_;
    const unsigned long long* seed = (const unsigned long long*)seed_;
    DDPM_REQUIRE(xt && x0_pred && t && sched && out, "reverse_step: null pointer");
    DDPM_REQUIRE(noise || seed, "reverse_step: need an external noise tensor or a device seed");
    DDPM_REQUIRE(N > 0 && chw > 0 && chw % 4 == 0, "reverse_step: C*H*W must be a positive multiple of 4 (got %lld)", chw);
    DDPM_REQUIRE(first_index % 4 == 0, "reverse_step: first_index must be a multiple of 4");
    hipLaunchKernelGGL(reverse_step_kernel, dim3(stream_grid((long long)N * chw / 4, 256)), dim3(256), 0, stream, xt, x0_pred, t, noise,
                       sched, out, N, chw, seed, stream_id, sigmoid_pred, (unsigned long long)(first_index / 4));
    return ddpm_check_launch("reverse_step");
}

// out[i] = the N(0,1) sample of flat element first_index + i of stream `stream_id` of the (*seed) sequence: the start tensor
// X_T of a sampling job (ddpm.py:152 of the reference: randn), keyed by GLOBAL element index so that a job sharded over
// several GPUs draws the same numbers whatever the number of GPUs.
__global__ __launch_bounds__(256) void fill_normal_kernel(float* __restrict__ out, long long n4, const unsigned long long* __restrict__ seed,
                                                           unsigned stream_id, unsigned long long q0) {
    const unsigned long long sd = *seed;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x)
        *(f32x4*)(out + q * 4) = philox_normal4(sd, stream_id, q0 + (unsigned long long)q);
}

extern "C" int ddpm_fill_normal(float* out, int64_t n, const uint64_t* seed, uint32_t stream_id, uint64_t first_index, ddpm_stream_t stream_) {
    DDPM_REQUIRE(out && seed, "fill_normal: null pointer");
    DDPM_REQUIRE(n > 0 && n % 4 == 0 && first_index % 4 == 0, "fill_normal: n and first_index must be multiples of 4");
    hipLaunchKernelGGL(fill_normal_kernel, dim3(stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream_, out, (long long)(n / 4),
                       (const unsigned long long*)seed, stream_id, (unsigned long long)(first_index / 4));
    return ddpm_check_launch("fill_normal");
}

// ------------------------------------------------------------------------------------------------
// ELBO: loss = mean_{n,c,h,w}( w_t[n] * l(pred, target) ), l = squared error or BCE-with-logits
// (reference losses/ELBO.py:61-76).  Forward reduces into one fp32 scalar (per-block partial sums
// combined with one atomic per block); backward writes dL/dpred scaled by an upstream scalar.
__device__ __forceinline__ float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void elbo_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const long long* __restrict__ t, const float* __restrict__ sched,
                                                        float* __restrict__ loss, int N, long long chw, int bce, int weighted) {
    __shared__ float part[4];
    const long long total4 = (long long)N * chw / 4;
    float acc = 0.f;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total4; q += (long long)gridDim.x * blockDim.x) {
        const long long e = q * 4;
        const int n = (int)(e / chw);
        const float w = weighted ? sched[(int)t[n] * SC_STRIDE + SC_W] : 1.0f;
        const f32x4 p = *(const f32x4*)(pred + e), y = *(const f32x4*)(target + e);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = p[j] - y[j];
            s += bce ? bce_logits(p[j], y[j]) : d * d;
        }
        acc += w * s;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) / (float)((double)N * (double)chw));
}

__global__ __launch_bounds__(256) void elbo_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const long long* __restrict__ t, const float* __restrict__ sched,
                                                        const float* __restrict__ gscale, float* __restrict__ gpred, int N,
                                                        long long chw, int bce, int weighted) {
    const long long total4 = (long long)N * chw / 4;
    const float up = (gscale ? *gscale : 1.0f) / (float)((double)N * (double)chw);
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < total4; q += (long long)gridDim.x * blockDim.x) {
        const long long e = q * 4;
        const int n = (int)(e / chw);
        const float w = (weighted ? sched[(int)t[n] * SC_STRIDE + SC_W] : 1.0f) * up;
        const f32x4 p = *(const f32x4*)(pred + e), y = *(const f32x4*)(target + e);
        f32x4 g;
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = w * (bce ? (1.0f / (1.0f + __expf(-p[j])) - y[j]) : 2.0f * (p[j] - y[j]));
        *(f32x4*)(gpred + e) = g;
    }
}

extern "C" int ddpm_elbo_fwd(const float* pred, const float* target, const int64_t* t_, const float* sched, float* loss, int N,
                             int64_t chw, int bce, int weighted, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long* t = (const long long*)t_;
    DDPM_REQUIRE(pred && target && t && sched && loss, "elbo_fwd: null pointer");
    DDPM_REQUIRE(N > 0 && chw > 0 && chw % 4 == 0, "elbo_fwd: C*H*W must be a positive multiple of 4 (got %lld)", chw);
    if (hipMemsetAsync(loss, 0, sizeof(float), stream) != hipSuccess) return ddpm_check_launch("elbo_fwd memset");
    hipLaunchKernelGGL(elbo_fwd_kernel, dim3(ddpm_deterministic() ? 1 : stream_grid((long long)N * chw / 4, 256)), dim3(256), 0, stream, pred,
                       target, t, sched, loss, N, chw, bce, weighted);
    return ddpm_check_launch("elbo_fwd");
}

extern "C" int ddpm_elbo_bwd(const float* pred, const float* target, const int64_t* t_, const float* sched, const float* gscale,
                             float* gpred, int N, int64_t chw, int bce, int weighted, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long* t = (const long long*)t_;
    DDPM_REQUIRE(pred && target && t && sched && gpred, "elbo_bwd: null pointer");
    DDPM_REQUIRE(N > 0 && chw > 0 && chw % 4 == 0, "elbo_bwd: C*H*W must be a positive multiple of 4 (got %lld)", chw);
    hipLaunchKernelGGL(elbo_bwd_kernel, dim3(stream_grid((long long)N * chw / 4, 256)), dim3(256), 0, stream, pred, target, t, sched,
                       gscale, gpred, N, chw, bce, weighted);
    return ddpm_check_launch("elbo_bwd");
}

// The squared-error ELBO of a tanh-headed denoiser and everything between it and the last convolution's weight / data
// gradients in one launch: loss (as elbo_fwd_kernel), dL/dpred (as elbo_bwd_kernel with no upstream scale), the tanh
// backward into the channel-padded NHWC bf16 gradient of the logits (as tanh_bwd_kernel) and that gradient's per-channel sum,
// the bias gradient of the last convolution (as colsum_kernel over the bf16 values).  Four dependent launches on a 1 MB
// tensor otherwise; same arithmetic and rounding points, so dz is bit-identical to the separate kernels' (the two sums are
// added in another order).  A thread owns one pixel with all its channels.
__global__ __launch_bounds__(256) void elbo_tanh_head_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                              const long long* __restrict__ t, const float* __restrict__ sched,
                                                              float* __restrict__ loss, bf16* __restrict__ dz, float* __restrict__ bias_grad,
                                                              int N, int C, int HW, int weighted) {
    __shared__ float part[4][17];
    const long long total = (long long)N * HW;
    const long long chw = (long long)C * HW;
    const float inv = 1.0f / (float)((double)N * (double)chw);
    float acc = 0.f, bs[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) bs[c] = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        const float wt = weighted ? sched[(int)t[n] * SC_STRIDE + SC_W] : 1.0f;
        const float w = wt * inv;
        bf16x8 lo, hi;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            bf16 o = (bf16)0.f;
            if (c < C) {
                const long long e = ((long long)n * C + c) * HW + p;
                const float yy = pred[e], d = yy - target[e];
                s += d * d;
                const float g = w * (2.0f * d);
                o = (bf16)(g * (1.0f - yy * yy));
                bs[c] += (float)o;
            }
            if (c < 8) lo[c & 7] = o; else hi[c & 7] = o;
        }
        acc += wt * s;
        *(bf16x8*)(dz + i * 16) = lo;
        *(bf16x8*)(dz + i * 16 + 8) = hi;
    }
    const int wave = threadIdx.x >> 6;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[wave][16] = acc;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < C) {
            const float v = wave_sum(bs[c]);
            if ((threadIdx.x & 63) == 0) part[wave][c] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0][16] + part[1][16] + part[2][16] + part[3][16]) * inv);
    if (threadIdx.x < C) atomicAdd(bias_grad + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

extern "C" int ddpm_elbo_tanh_head(const float* pred, const float* target, const int64_t* t_, const float* sched, float* loss, void* dz,
                                   float* bias_grad, int N, int C, int HW, int weighted, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(pred && target && t_ && sched && loss && dz && bias_grad, "elbo_tanh_head: null pointer");
    DDPM_REQUIRE(N > 0 && C > 0 && C <= 16 && HW > 0, "elbo_tanh_head: 1..16 image channels (got %d)", C);
    if (hipMemsetAsync(loss, 0, sizeof(float), stream) != hipSuccess) return ddpm_check_launch("elbo_tanh_head memset");
    hipLaunchKernelGGL(elbo_tanh_head_kernel, dim3(ddpm_deterministic() ? 1 : stream_grid((long long)N * HW, 256)), dim3(256), 0, stream, pred, target,
                       (const long long*)t_, sched, loss, (bf16*)dz, bias_grad, N, C, HW, weighted);
    return ddpm_check_launch("elbo_tanh_head");
}

// ------------------------------------------------------------------------------------------------
// Sinusoidal time embedding (reference constituent_blocks.py:261-279; time normalised as in
// unets.py:121): out[n] = [sin(t/Tn * f_i), cos(t/Tn * f_i)], f_i = exp(-i ln(1e4)/(half-1)).
__global__ void time_embedding_kernel(const long long* __restrict__ t, float inv_norm, int dim, bf16* __restrict__ out, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (i >= N * half) return;
    const int n = i / half, k = i - n * half;
    const float pre = 9.210340371976184f / (float)(half - 1);
    const float f = expf(-pre * (float)k);
    const float a = ((float)t[n] * inv_norm) * f;
    out[n * dim + k] = (bf16)sinf(a);
    out[n * dim + half + k] = (bf16)cosf(a);
}

extern "C" int ddpm_time_embedding(const int64_t* t_, float time_norm, int dim, void* out, int N, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const long long* t = (const long long*)t_;
    DDPM_REQUIRE(t && out && N > 0 && dim >= 4 && dim % 2 == 0, "time_embedding: bad arguments");
    const int total = N * (dim / 2);
    hipLaunchKernelGGL(time_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, t, 1.0f / time_norm, dim, (bf16*)out, N);
    return ddpm_check_launch("time_embedding");
}

// ------------------------------------------------------------------------------------------------
// Layout bridges.  The UNet works on NHWC bf16 internally; the module boundary is NCHW fp32.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, bf16* __restrict__ out, int N, int C, int HW,
                                                            int cpad) {
    const long long total = (long long)N * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        if (cpad == 16) {             // images: the pixel's 32 bytes (channels + zero padding) as two vector stores
            bf16x8 lo, hi;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                lo[c] = c < C ? (bf16)x[((long long)n * C + c) * HW + p] : (bf16)0.f;
                hi[c] = c + 8 < C ? (bf16)x[((long long)n * C + c + 8) * HW + p] : (bf16)0.f;
            }
            *(bf16x8*)(out + i * 16) = lo;
            *(bf16x8*)(out + i * 16 + 8) = hi;
            continue;
        }
        for (int c = 0; c < cpad; ++c) out[i * cpad + c] = c < C ? (bf16)x[((long long)n * C + c) * HW + p] : (bf16)0.f;
    }
}

extern "C" int ddpm_nchw_f32_to_nhwc_bf16(const float* x, void* out, int N, int C, int HW, int cpad, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && out && N > 0 && C > 0 && HW > 0 && cpad >= C, "nchw_to_nhwc: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_grid((long long)N * HW, 256)), dim3(256), 0, stream, x, (bf16*)out, N, C, HW, cpad);
    return ddpm_check_launch("nchw_to_nhwc");
}

// dz = dy * (1 - y^2) for y = tanh(z): turns the gradient arriving at the UNet output (NCHW fp32)
// into the channel-padded NHWC bf16 gradient of the last convolution's logits.
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, bf16* __restrict__ dz,
                                                        int N, int C, int HW, int cpad) {
    const long long total = (long long)N * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (long long)n * HW);
        if (cpad == 16) {             // images: the pixel's 32 bytes as two vector stores
            bf16x8 lo, hi;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float g = 0.f;
                if (c < C) {
                    const long long s = ((long long)n * C + c) * HW + p;
                    const float yy = y[s];
                    g = dy[s] * (1.0f - yy * yy);
                }
                if (c < 8) lo[c & 7] = (bf16)g; else hi[c & 7] = (bf16)g;
            }
            *(bf16x8*)(dz + i * 16) = lo;
            *(bf16x8*)(dz + i * 16 + 8) = hi;
            continue;
        }
        for (int c = 0; c < cpad; ++c) {
            float g = 0.f;
            if (c < C) {
                const long long s = ((long long)n * C + c) * HW + p;
                const float yy = y[s];
                g = dy[s] * (1.0f - yy * yy);
            }
            dz[i * cpad + c] = (bf16)g;
        }
    }
}

extern "C" int ddpm_tanh_bwd_nchw_to_nhwc(const float* dy, const float* y, void* dz, int N, int C, int HW, int cpad, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(dy && y && dz && N > 0 && C > 0 && HW > 0 && cpad >= C, "tanh_bwd: bad arguments");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(stream_grid((long long)N * HW, 256)), dim3(256), 0, stream, dy, y, (bf16*)dz, N, C, HW, cpad);
    return ddpm_check_launch("tanh_bwd");
}

// NHWC bf16 -> NCHW fp32 (gradient w.r.t. the UNet input, when a caller asks for it).
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const bf16* __restrict__ x, float* __restrict__ out, int N, int C, int HW,
                                                            int cpad) {
    const long long total = (long long)N * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const long long nc = i / HW;
        const int c = (int)(nc % C);
        const int n = (int)(nc / C);
        out[i] = (float)x[((long long)n * HW + p) * cpad + c];
    }
}

extern "C" int ddpm_nhwc_bf16_to_nchw_f32(const void* x, float* out, int N, int C, int HW, int cpad, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && out && N > 0 && C > 0 && HW > 0 && cpad >= C, "nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(stream_grid((long long)N * C * HW, 256)), dim3(256), 0, stream, (const bf16*)x, out, N, C,
                       HW, cpad);
    return ddpm_check_launch("nhwc_to_nchw");
}

// Advance a device-resident 64-bit seed by one (so a replayed graph draws fresh randomness each step).
// Backward of a bare activation between a convolution and its consumer (conv -> act units without normalisation, e.g. the
// stem, conv_block.py:19-54 there with norm = None): dx = dy * act'(x), plus the column sums of dx (the convolution's
// bias gradient) as one partial row per workgroup: part[(n * P + p)][C], image n cut into P pixel ranges.  Many
// workgroups per image: the tensors of these units are the largest of the network and the work is pure streaming.
// A: the activation as a compile-time code (gelu, silu), -1 = the runtime code (a runtime switch per element is a scalar
// branch per element and keeps the arithmetic from being packed).
template <int A>
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                                       float* __restrict__ part, int HW, int C, int P, int act) {
    __shared__ float s_red[4 * 64];
    const int tid = threadIdx.x;
    const int vps = C / 8;                           // power of two <= 8 (host-checked)
    const int n = blockIdx.x / P, p = blockIdx.x - n * P;
    const int rows = HW / P;
    const int cv = tid % vps, prow = tid / vps, pstep = 256 / vps;
    const long long base = ((long long)n * HW + (long long)p * rows) * C + cv * 8;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = 0.f;
#pragma unroll 4
    for (int r = prow; r < rows; r += pstep) {
        const long long e = base + (long long)r * C;
        const bf16x8 xv = *(const bf16x8*)(x + e), dv = *(const bf16x8*)(dy + e);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = (float)dv[j] * act_bwd((float)xv[j], A >= 0 ? A : act);
            sum[j] += d;
            o[j] = (bf16)d;
        }
        *(bf16x8*)(dx + e) = o;
    }
    block_reduce_cols<256>(sum, vps, s_red, tid);
    if (tid < vps) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[(long long)blockIdx.x * C + tid * 8 + j] = sum[j];
    }
}

extern "C" int ddpm_act_backward(const void* x, const void* dy, void* dx, float* part, int N, int HW, int C, int P, int act,
                                 ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && dy && dx && part && N > 0 && HW > 0, "act_backward: bad arguments");
    DDPM_REQUIRE(C == 8 || C == 16 || C == 32 || C == 64, "act_backward: C must be 8, 16, 32 or 64 (got %d)", C);
    DDPM_REQUIRE(P > 0 && HW % P == 0 && (long long)N * P < (1ll << 31), "act_backward: P (%d) must divide H*W (%d)", P, HW);
    DDPM_REQUIRE(act >= 0 && act <= ACT_LAST && !ACT_HAS_BETA(act), "act_backward: unknown activation %d (swish / hardswish go through ddpm_gn_act_backward)", act);
    auto k = act == ACT_GELU ? act_bwd_kernel<ACT_GELU> : act == ACT_SILU ? act_bwd_kernel<ACT_SILU> : act_bwd_kernel<-1>;
    hipLaunchKernelGGL(k, dim3((unsigned)(N * P)), dim3(256), 0, stream, (const bf16*)x, (const bf16*)dy, (bf16*)dx, part, HW, C, P, act);
    return ddpm_check_launch("act_backward");
}

// t ~ U{1..T} per image (base_trainer.py:202 there draws it with torch.randint): Philox word i of the (seed, stream) sequence,
// mapped to the range by multiply-high.  Drawing it here keeps the framework's generator (and the bookkeeping kernels
// a captured graph needs for it) out of the training step.
__global__ __launch_bounds__(256) void sample_timesteps_kernel(long long* __restrict__ t, int N, int T,
                                                                const unsigned long long* __restrict__ seed, unsigned stream_id) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= N) return;
    const unsigned long long sd = *seed;
    const Philox4 r = philox4x32_10((uint32_t)q, 0u, stream_id, 0x3c6ef372u, (uint32_t)sd, (uint32_t)(sd >> 32));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (q * 4 + j < N) t[q * 4 + j] = 1 + (long long)(((unsigned long long)w[j] * (unsigned)T) >> 32);
}

extern "C" int ddpm_sample_timesteps(int64_t* t, int N, int T, const uint64_t* seed, uint32_t stream_id, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(t && seed && N > 0 && T > 0, "sample_timesteps: bad arguments");
    hipLaunchKernelGGL(sample_timesteps_kernel, dim3((N + 1023) / 1024), dim3(256), 0, stream, (long long*)t, N, T,
                       (const unsigned long long*)seed, stream_id);
    return ddpm_check_launch("sample_timesteps");
}

// ------------------------------------------------------------------------------------------------
// The head of a training step in one launch: per-image timestep draw (as sample_timesteps_kernel), forward noising
// (as q_sample_kernel) written straight into the channel-padded NHWC bf16 tensor the stem convolution reads, the
// sinusoidal time embedding (as time_embedding_kernel), and the seed advance (as bump_seed_kernel) by the last workgroup
// to finish.  Five dependent tiny launches otherwise open every step on an idle queue.  Same draws, same arithmetic and
// the same rounding points as the separate kernels.
__device__ __forceinline__ long long draw_timestep(unsigned long long sd, int n, int T) {
    const Philox4 r = philox4x32_10((uint32_t)(n >> 2), 0u, 3u, 0x3c6ef372u, (uint32_t)sd, (uint32_t)(sd >> 32));
    const uint32_t w = (n & 3) == 0 ? r.x : (n & 3) == 1 ? r.y : (n & 3) == 2 ? r.z : r.w;
    return 1 + (long long)(((unsigned long long)w * (unsigned)T) >> 32);
}

__global__ __launch_bounds__(256) void train_head_kernel(const float* __restrict__ x0, const long long* __restrict__ t_in,
                                                          const float* __restrict__ noise, const float* __restrict__ sched,
                                                          long long* __restrict__ t_out, bf16* __restrict__ xt_nhwc, int cpad,
                                                          bf16* __restrict__ sin_out, float inv_norm, int dim, int N, int C, int HW, int T,
                                                          unsigned long long* __restrict__ seed, unsigned* __restrict__ done) {
    const unsigned long long sd = *seed;
    const long long chw = (long long)C * HW;
    const long long total4 = (long long)N * chw / 4;
    const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x, gstride = (long long)gridDim.x * blockDim.x;
    if (cpad == 16) {
        // images (C <= 16): a thread owns four pixels with all their channels and stores each pixel's 32 bytes (channels
        // and zero padding) as two vectors; per-channel scalar stores would be 64 two-byte stores per thread
        const int hw4 = HW / 4;
        for (long long it = gid; it < (long long)N * hw4; it += gstride) {
            const int n = (int)(it / hw4);
            const int p = (int)(it - (long long)n * hw4) * 4;
            const int ti = (int)(t_in ? t_in[n] : draw_timestep(sd, n, T));
            const float a = sched[ti * SC_STRIDE + SC_SQRT_AB], b = sched[ti * SC_STRIDE + SC_SQRT_1MAB];
            bf16x8 lo[4], hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) { lo[j][k] = (bf16)0.f; hi[j][k] = (bf16)0.f; }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c < C) {
                    const long long e = ((long long)n * C + c) * HW + p;
                    const f32x4 x = *(const f32x4*)(x0 + e);
                    const f32x4 z = noise ? *(const f32x4*)(noise + e) : philox_normal4(sd, 1u, (unsigned long long)(e >> 2));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16 o = (bf16)(a * x[j] + b * z[j]);      // two roundings + one, as q_sample_kernel (no FMA)
                        if (c < 8) lo[j][c & 7] = o; else hi[j][c & 7] = o;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16* px = xt_nhwc + ((long long)n * HW + p + j) * 16;
                *(bf16x8*)px = lo[j];
                *(bf16x8*)(px + 8) = hi[j];
            }
        }
    } else {
        for (long long q = gid; q < total4; q += gstride) {
            const long long e = q * 4;
            const int n = (int)(e / chw);
            const int ti = (int)(t_in ? t_in[n] : draw_timestep(sd, n, T));
            const float a = sched[ti * SC_STRIDE + SC_SQRT_AB], b = sched[ti * SC_STRIDE + SC_SQRT_1MAB];
            const f32x4 x = *(const f32x4*)(x0 + e);
            const f32x4 z = noise ? *(const f32x4*)(noise + e) : philox_normal4(sd, 1u, (unsigned long long)q);
            const long long r = e - (long long)n * chw;
            const int c = (int)(r / HW);
            const int p = (int)(r - (long long)c * HW);          // HW % 4 == 0: the four elements share n and c
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float o = a * x[j] + b * z[j];
                bf16* px = xt_nhwc + ((long long)n * HW + p + j) * cpad;
                px[c] = (bf16)o;
                if (c == 0)
                    for (int cc = C; cc < cpad; ++cc) px[cc] = (bf16)0.f;
            }
        }
    }
    const int half = dim / 2;
    for (long long i = gid; i < (long long)N * half; i += gstride) {
        const int n = (int)(i / half), k = (int)(i - (long long)n * half);
        const long long tn = t_in ? t_in[n] : draw_timestep(sd, n, T);
        if (k == 0) t_out[n] = tn;
        const float pre = 9.210340371976184f / (float)(half - 1);
        const float f = expf(-pre * (float)k);
        const float ang = ((float)tn * inv_norm) * f;
        sin_out[n * dim + k] = (bf16)sinf(ang);
        sin_out[n * dim + half + k] = (bf16)cosf(ang);
    }
    // the seed advances once every workgroup has read it: the last one to arrive here does it
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(done, 1u) == gridDim.x - 1) {
            *seed = sd + 0x9E3779B97F4A7C15ull;
            *done = 0u;
        }
    }
}

extern "C" int ddpm_train_head(const float* x0, const int64_t* t_in, const float* noise, const float* sched, int64_t* t_out, void* xt_nhwc,
                               int cpad, void* sin_out, float time_norm, int t_dim, int N, int C, int HW, int T, uint64_t* seed,
                               uint32_t* done, ddpm_stream_t stream_) {
    DDPM_REQUIRE(x0 && sched && t_out && xt_nhwc && sin_out && seed && done, "train_head: null pointer");
    DDPM_REQUIRE(N > 0 && C > 0 && HW > 0 && HW % 4 == 0 && cpad >= C && T > 0, "train_head: H*W must be a positive multiple of 4, cpad >= C");
    DDPM_REQUIRE(t_dim >= 4 && t_dim % 2 == 0, "train_head: bad embedding width %d", t_dim);
    const long long total4 = cpad == 16 ? (long long)N * HW / 4 : (long long)N * C * HW / 4;
    hipLaunchKernelGGL(train_head_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream_, x0, (const long long*)t_in, noise,
                       sched, (long long*)t_out, (bf16*)xt_nhwc, cpad, (bf16*)sin_out, 1.0f / time_norm, t_dim, N, C, HW, T,
                       (unsigned long long*)seed, (unsigned*)done);
    return ddpm_check_launch("train_head");
}

__global__ void bump_seed_kernel(unsigned long long* seed) { *seed += 0x9E3779B97F4A7C15ull; }

extern "C" int ddpm_bump_seed(uint64_t* seed_, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    unsigned long long* seed = (unsigned long long*)seed_;
    DDPM_REQUIRE(seed, "bump_seed: null pointer");
    hipLaunchKernelGGL(bump_seed_kernel, dim3(1), dim3(1), 0, stream, seed);
    return ddpm_check_launch("bump_seed");
}

// Range check of a timestep vector without stalling the host: safe[i] = t[i] clamped into [lo, hi] (what the kernels that
// index the per-timestep table then read), and for an out-of-range value flag[0] = 1, flag[1] = that value (saturated to
// 32 bits).  flag lives in host-pinned memory: the host looks at it whenever it likes, never waiting for the device.
__global__ __launch_bounds__(256) void check_timesteps_kernel(const long long* __restrict__ t, long long* __restrict__ safe, int n, int lo,
                                                               int hi, int* __restrict__ flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long v = t[i];
    if (v < lo || v > hi) {
        flag[1] = v < -2147483647ll ? -2147483647 : v > 2147483647ll ? 2147483647 : (int)v;
        flag[0] = 1;
    }
    safe[i] = v < lo ? lo : v > hi ? hi : v;
}

extern "C" int ddpm_check_timesteps(const int64_t* t, int64_t* safe, int n, int lo, int hi, int32_t* flag, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(t && safe && flag && n > 0 && lo <= hi, "check_timesteps: bad arguments");
    hipLaunchKernelGGL(check_timesteps_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const long long*)t, (long long*)safe, n, lo, hi,
                       (int*)flag);
    return ddpm_check_launch("check_timesteps");
}

// ------------------------------------------------------------------------------------------------
// Column sums of an NHWC bf16 matrix [rows][cs] -> fp32 [C] (+=): the bias gradient of a convolution whose
// output gradient is not produced by a GroupNorm backward (the image head, the time-embedding MLPs).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, float* __restrict__ out, long long rows, int C, int cs,
                                                      int rows_per_block, int cvs, int fixed_order) {
    // fixed_order (one workgroup, reproducible sums): the four waves' partial sums meet in LDS and are added in wave order
    __shared__ float s_wave[4][64][8];
    const int cv = threadIdx.x % cvs, prow = threadIdx.x / cvs, pstep = 256 / cvs;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (prow < pstep && cv * 8 < C) {
        for (long long r = r0 + prow; r < r1; r += pstep) {
            const bf16x8 v = *(const bf16x8*)(x + r * cs + cv * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
        }
    }
    const int lane = threadIdx.x & 63;
    const int lo = cvs < 64 ? cvs : 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = s[j];
        for (int o = 32; o >= lo; o >>= 1) v += __shfl_xor(v, o, 64);
        const int c = cv * 8 + j;
        if (fixed_order) {
            if (lane < lo) s_wave[threadIdx.x >> 6][lane][j] = v;
        } else if (lane < lo && c < C) {
            atomicAdd(out + c, v);
        }
    }
    if (fixed_order) {
        __syncthreads();
        // cvs <= 64: wave w holds the rows (w * 64 + lane) / cvs of every pass, i.e. all four waves hold partial sums of every column
        if (threadIdx.x < lo) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = threadIdx.x * 8 + j;
                if (c < C) out[c] += ((s_wave[0][threadIdx.x][j] + s_wave[1][threadIdx.x][j]) + s_wave[2][threadIdx.x][j]) + s_wave[3][threadIdx.x][j];
            }
        }
    }
}

extern "C" int ddpm_colsum(const void* x, float* out, int64_t rows, int C, int cs, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && out && rows > 0 && C > 0 && cs >= C && cs % 8 == 0, "colsum: bad arguments");
    // vector columns per row handled by one pass: the power of two covering C (the first C channels of each row; the row
    // stride cs is free, so a channel range of a wider packed tensor can be summed by offsetting x)
    int cvs = 1;
    while (cvs * 8 < C) cvs <<= 1;
    DDPM_REQUIRE(cvs <= 64 && C % 8 == 0 || cvs * 8 <= cs, "colsum: C = %d must be at most 512 (and a multiple of 8 unless the row is padded)", C);
    DDPM_REQUIRE(cvs <= 64, "colsum: C = %d must be at most 512", C);
    long long blocks = (rows * cvs + 256 * 16 - 1) / (256 * 16);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1 || ddpm_deterministic()) blocks = 1;
    DDPM_REQUIRE(rows / blocks < (1ll << 31), "colsum: too many rows for one workgroup");
    const int rpb = (int)((rows + blocks - 1) / blocks);
    blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)x, out, (long long)rows, C, cs, rpb, cvs,
                       (int)(blocks == 1 && ddpm_deterministic()));
    return ddpm_check_launch("colsum");
}

// out[n][c] = sum_p x[n][p][c]  (fp32, overwritten): gradient of the broadcast time-embedding add.
// One workgroup per (image, 64-channel slab).
__global__ __launch_bounds__(256) void spatial_sum_kernel(const bf16* __restrict__ x, float* __restrict__ out, int HW, int C, int SC) {
    __shared__ float s_red[4 * 64];
    const int slabs = C / SC;
    const int n = blockIdx.x / slabs, c0 = (blockIdx.x - n * slabs) * SC;
    const int vps = SC / 8;
    const int cv = threadIdx.x % vps, prow = threadIdx.x / vps, pstep = 256 / vps;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    for (int p = prow; p < HW; p += pstep) {
        const bf16x8 v = *(const bf16x8*)(x + ((long long)n * HW + p) * C + c0 + cv * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = s[j];
        for (int o = 32; o >= vps; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane < vps) s_red[wave * 64 + lane * 8 + j] = v;
    }
    __syncthreads();
    if (threadIdx.x < SC) out[(long long)n * C + c0 + threadIdx.x] = s_red[threadIdx.x] + s_red[64 + threadIdx.x] + s_red[128 + threadIdx.x] + s_red[192 + threadIdx.x];
}

extern "C" int ddpm_spatial_sum(const void* x, float* out, int N, int HW, int C, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "spatial_sum: bad arguments");
    int sc = C < 64 ? C : 64;
    while (C % sc) sc -= 8;
    DDPM_REQUIRE(sc == 8 || sc == 16 || sc == 32 || sc == 64, "spatial_sum: unsupported channel count %d", C);
    hipLaunchKernelGGL(spatial_sum_kernel, dim3(N * (C / sc)), dim3(256), 0, stream, (const bf16*)x, out, HW, C, sc);
    return ddpm_check_launch("spatial_sum");
}

// Dropout keep bits for all sites of a step in one launch.  Byte b of site s holds the keep decisions of the
// 8 consecutive elements 8b..8b+7 of that site's tensor.  A thread writes 16 bytes (128 elements) per round with one
// site look-up; only a 16-byte run that crosses into the next site (or the end of the buffer) looks its words up one by one.
__global__ __launch_bounds__(256) void dropout_masks_kernel(unsigned char* __restrict__ bytes, const long long* __restrict__ offsets,
                                                             int nsites, long long total_bytes, unsigned thresh,
                                                             const unsigned long long* __restrict__ seed) {
    const unsigned long long sd = *seed;
    auto site_of = [&](long long i) {           // last site whose offset <= i
        int lo = 0, hi = nsites - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offsets[mid] <= i) lo = mid; else hi = mid - 1;
        }
        return lo;
    };
    // every site's byte range starts and ends on a multiple of 4, so a 4-byte word never straddles two sites
    for (long long i = (blockIdx.x * 256ll + threadIdx.x) * 16; i < total_bytes; i += (long long)gridDim.x * 256 * 16) {
        const int lo = site_of(i);
        const long long end = lo + 1 < nsites ? offsets[lo + 1] : total_bytes;
        if (i + 16 <= end) {
            const long long rel = i - offsets[lo];
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[k] = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) w[k] |= dropout_keep8(sd, (unsigned)lo, (unsigned long long)(rel + 4 * k + b), thresh) << (8 * b);
            }
            *(uint4*)(bytes + i) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (long long j = i; j < i + 16 && j < total_bytes; j += 4) {
                const int s = site_of(j);
                const long long rel = j - offsets[s];
                unsigned word = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) word |= dropout_keep8(sd, (unsigned)s, (unsigned long long)(rel + b), thresh) << (8 * b);
                *(unsigned*)(bytes + j) = word;
            }
        }
    }
}

extern "C" int ddpm_dropout_masks(uint8_t* bytes, const int64_t* offsets, int nsites, int64_t total_bytes, float keep,
                                  const uint64_t* seed, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(bytes && offsets && seed && nsites > 0 && total_bytes > 0 && total_bytes % 4 == 0, "dropout_masks: bad arguments");
    DDPM_REQUIRE(keep > 0.f && keep <= 1.f, "dropout_masks: keep probability %f out of range", (double)keep);
    DDPM_REQUIRE(((uintptr_t)bytes & 15) == 0, "dropout_masks: the bit buffer must be 16-byte aligned");
    const unsigned thresh = (unsigned)(keep * 65536.0f + 0.5f);
    // At most one workgroup per CU: the kernel is ALU bound (70 us with two per CU against 62 us with every wave slot taken, for
    // 22 MB of bits), and it usually runs on a side stream beside latency-bound kernels whose workgroups must still find free
    // wave slots (a full-size grid delayed the first kernels of a training step by ~50 us; one workgroup per CU instead of two
    // takes twice as long beside them and costs the step 10 us less; half of that again costs more).  DDPM_MASK_GRID: a
    // measurement switch for this bound.
    int mgrid = stream_grid((total_bytes + 15) / 16, 256);
    static const int cap = [] { const char* e = getenv("DDPM_MASK_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    if (mgrid > cap) mgrid = cap;
    hipLaunchKernelGGL(dropout_masks_kernel, dim3(mgrid), dim3(256), 0, stream, bytes,
                       (const long long*)offsets, nsites, (long long)total_bytes, thresh, (const unsigned long long*)seed);
    return ddpm_check_launch("dropout_masks");
}

// ------------------------------------------------------------------------------------------------
// Second stage of the deterministic two-stage reductions (GroupNorm / bias / depthwise parameter gradients):
// for each table entry, dst[dst_off + j] += sum_r src[src_off + r*row_stride + j].  One launch for all layers.
// Workgroup = 64 columns x 4 row lanes.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           const ddpm_reduce_entry* __restrict__ entries) {
    __shared__ float s[4][64];
    const ddpm_reduce_entry e = entries[blockIdx.y];
    if (e.scalar) {                                     // a contiguous run -> one number (fixed order: deterministic)
        if (blockIdx.x) return;
        float acc = 0.f;
        const float* p = src + e.src_off;
#pragma unroll 4
        for (int i = threadIdx.x; i < e.rows; i += 256) acc += p[i];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) s[0][threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) dst[e.dst_off] += (s[0][0] + s[0][1]) + (s[0][2] + s[0][3]);
        return;
    }
    if (blockIdx.x * 64 >= e.width) return;
    if (!((e.src_off | e.dst_off | e.row_stride | e.width) & 3)) {
        // everything a multiple of 4 floats (all but a few odd-sized tensors): 16 column quads x 16 row lanes, so a thread
        // keeps 16-byte loads of up to four rows in flight and the row loop is a quarter as long
        __shared__ f32x4 s4[16][16];
        const int cq = threadIdx.x & 15, rl16 = threadIdx.x >> 4;
        const int c4 = blockIdx.x * 64 + cq * 4;
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        if (c4 < e.width) {
            const float* p = src + e.src_off + c4;
#pragma unroll 4
            for (int r = rl16; r < e.rows; r += 16) acc4 += *(const f32x4*)(p + (long long)r * e.row_stride);
        }
        s4[rl16][cq] = acc4;
        __syncthreads();
        if (rl16 == 0 && c4 < e.width) {
            f32x4 t = s4[0][cq];
#pragma unroll
            for (int k = 1; k < 16; ++k) t += s4[k][cq];
            f32x4* d = (f32x4*)(dst + e.dst_off + c4);
            *d = *d + t;
        }
        return;
    }
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < e.width) {
        const float* p = src + e.src_off + col;
#pragma unroll 4
        for (int r = rl; r < e.rows; r += 4) acc += p[(long long)r * e.row_stride];
    }
    s[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl == 0 && col < e.width) dst[e.dst_off + col] += s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

extern "C" int ddpm_reduce_rows(const float* src, float* dst, const ddpm_reduce_entry* entries, int n_entries, int max_width,
                                ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(src && dst && entries && n_entries > 0 && max_width > 0, "reduce_rows: bad arguments");
    DDPM_REQUIRE(n_entries <= 65535, "reduce_rows: too many entries");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((max_width + 63) / 64, n_entries), dim3(256), 0, stream, src, dst, entries);
    return ddpm_check_launch("reduce_rows");
}

// ------------------------------------------------------------------------------------------------
// Input pipeline of the reference's dataset loaders on the GPU (datasets/mnist.py:26-39, datasets/flowers102.py:28-38
// there): transforms.Resize (PIL bilinear: a horizontal then a vertical pass in 22-bit fixed point, each rounded to 8
// bits) -> ToTensor (/255) -> x*2-1 -> channel replication, plus an optional per-image horizontal flip.  The coefficient
// tables come from the host (same arithmetic as PIL's precompute_coeffs), so the result is bit-identical to the
// reference's CPU transforms.  One thread per output pixel (all channels).
__device__ __forceinline__ int clip8_fixed(int v) {
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void prepare_images_kernel(const unsigned char* __restrict__ src, int N, int H0, int W0, int C0,
                                                             float* __restrict__ dst, int C, int H, int W,
                                                             const int* __restrict__ hcoef, const int* __restrict__ hbounds, int hk,
                                                             const int* __restrict__ vcoef, const int* __restrict__ vbounds, int vk,
                                                             const unsigned char* __restrict__ flip) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * H * W) return;
    const int x = (int)(idx % W), y = (int)((idx / W) % H), n = (int)(idx / ((long long)W * H));
    const int xs = (flip && flip[n]) ? W - 1 - x : x;          // flipping the output = sampling the mirrored column
    const unsigned char* img = src + (long long)n * H0 * W0 * C0;
    const int ymin = vbounds ? vbounds[y * 2] : y, ycnt = vbounds ? vbounds[y * 2 + 1] : 1;
    const int xmin = hbounds ? hbounds[xs * 2] : xs, xcnt = hbounds ? hbounds[xs * 2 + 1] : 1;
    for (int c0 = 0; c0 < C0; ++c0) {
        int acc_v = 1 << 21;
        int u = 0;
        for (int j = 0; j < ycnt; ++j) {
            const unsigned char* row = img + ((long long)(ymin + j) * W0) * C0 + c0;
            int hval;
            if (hbounds) {
                int acc_h = 1 << 21;
                for (int i = 0; i < xcnt; ++i) acc_h += (int)row[(xmin + i) * C0] * hcoef[xs * hk + i];
                hval = clip8_fixed(acc_h);
            } else {
                hval = row[xmin * C0];
            }
            if (vbounds) acc_v += hval * vcoef[y * vk + j];
            else u = hval;
        }
        if (vbounds) u = clip8_fixed(acc_v);
        const float v = (float)u / 255.0f * 2.0f - 1.0f;
        if (C0 == C) {
            dst[(((long long)n * C + c0) * H + y) * W + x] = v;
        } else {                                               // one source channel replicated (img.expand(3, -1, -1))
            for (int c = 0; c < C; ++c) dst[(((long long)n * C + c) * H + y) * W + x] = v;
        }
    }
}

extern "C" int ddpm_prepare_images(const uint8_t* src, int N, int H0, int W0, int C0, float* dst, int C, int H, int W, const int32_t* hcoef,
                                   const int32_t* hbounds, int hk, const int32_t* vcoef, const int32_t* vbounds, int vk, const uint8_t* flip,
                                   ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(src && dst && N > 0 && H0 > 0 && W0 > 0 && H > 0 && W > 0, "prepare_images: bad arguments");
    DDPM_REQUIRE(C0 == C || C0 == 1, "prepare_images: %d source channels cannot fill %d output channels", C0, C);
    DDPM_REQUIRE((hbounds != nullptr) == (hcoef != nullptr) && (vbounds != nullptr) == (vcoef != nullptr), "prepare_images: coefficient tables come in pairs");
    DDPM_REQUIRE((hbounds || W0 == W) && (vbounds || H0 == H), "prepare_images: sizes differ but no resampling tables were given");
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(prepare_images_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, N, H0, W0, C0, dst, C, H, W,
                       hcoef, hbounds, hk, vcoef, vbounds, vk, flip);
    return ddpm_check_launch("prepare_images");
}
