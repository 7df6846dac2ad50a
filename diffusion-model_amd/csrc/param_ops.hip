// Multi-tensor parameter plumbing, one launch each for the whole network:
//   - remap: fp32 torch-layout parameters -> bf16/fp32 kernel-layout copies (forward and data-gradient forms of every
//     convolution weight, tap-major depthwise weights), and fp32 kernel-layout gradients -> fp32 torch-layout gradients;
//   - Adam (+ optional EMA) over one flat parameter buffer.
// The 350 parameter tensors of the denoiser are far too small to deserve a kernel each.
#include "common.h"
#include "ddpm_hip.h"

// A[s][r][k] (dense, dims S x R x K, zero outside r < Rv, k < Kv)  <->  B[r*a_r + k*a_k + off_s[s]]
// Work is cut into chunks of REMAP_CHUNK elements of A; chunks[i] = {entry index, first element / REMAP_CHUNK}, so
// every workgroup has the same amount of work whatever the mix of tensor sizes.
#define REMAP_CHUNK 2048
__global__ __launch_bounds__(256) void remap_kernel(const ddpm_remap_entry* __restrict__ entries, const int2* __restrict__ chunks,
                                                     int to_a) {
    const int2 ch = chunks[blockIdx.x];
    const ddpm_remap_entry* __restrict__ ep = entries + ch.x;      // off_s is indexed per element: a by-value copy of the entry would live in scratch memory
    struct { int S, R, K, Rv, Kv, a_r, a_k, a_bf16; void* a; void* b; } e = {ep->S, ep->R, ep->K, ep->Rv, ep->Kv, ep->a_r, ep->a_k, ep->a_bf16, (void*)ep->a, (void*)ep->b};
    // one tensor has far fewer than 2^31 elements (checked by the host side when the table is built): 32-bit index
    // arithmetic, whose divisions cost a fraction of the 64-bit ones
    const unsigned total = (unsigned)e.S * (unsigned)e.R * (unsigned)e.K;
    const unsigned i0 = (unsigned)ch.y * REMAP_CHUNK;
    const unsigned K = (unsigned)e.K, R = (unsigned)e.R;
    // eight consecutive k per thread where K is a multiple of 8 (A's base is 16-byte aligned: weight copies are tensors of
    // their own, gradient slots start at multiples of 4 floats): the index arithmetic is paid once per 8 elements and A is
    // moved as one 16- or 32-byte vector
    const unsigned i = i0 + threadIdx.x * 8;
    if (i >= total) return;
    if (K % 8) {                                   // small odd-sized tensors (a 1- or 3-channel bias): element by element
        for (unsigned q = i; q < i + 8 && q < total; ++q) {
            const unsigned sr = q / K;
            const int k = (int)(q - sr * K);
            const int s = (int)(sr / R), r = (int)(sr - (unsigned)s * R);
            const bool valid = r < e.Rv && k < e.Kv;
            const int bi = r * e.a_r + k * e.a_k + ep->off_s[s];
            if (to_a) {
                const float v = valid ? ((const float*)e.b)[bi] : 0.f;
                if (e.a_bf16) ((bf16*)e.a)[q] = (bf16)v;
                else ((float*)e.a)[q] = v;
            } else if (valid) {
                ((float*)e.b)[bi] = e.a_bf16 ? (float)((const bf16*)e.a)[q] : ((const float*)e.a)[q];
            }
        }
        return;
    }
    const unsigned sr = i / K;
    const int k0 = (int)(i - sr * K);
    const int s = (int)(sr / R), r = (int)(sr - (unsigned)s * R);
    const int bi = r * e.a_r + k0 * e.a_k + ep->off_s[s];
    float* const b = (float*)e.b;
    if (to_a) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (r < e.Rv && k0 + j < e.Kv) ? b[bi + j * e.a_k] : 0.f;
        if (e.a_bf16) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
            *(bf16x8*)((bf16*)e.a + i) = o;
        } else {
            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            *(f32x4*)((float*)e.a + i) = o0; *(f32x4*)((float*)e.a + i + 4) = o1;
        }
    } else if (r < e.Rv) {
        float v[8];
        if (e.a_bf16) {
            const bf16x8 x = *(const bf16x8*)((const bf16*)e.a + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
        } else {
            const f32x4 x0 = *(const f32x4*)((const float*)e.a + i), x1 = *(const f32x4*)((const float*)e.a + i + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < e.Kv) b[bi + j * e.a_k] = v[j];
    }
}

extern "C" int ddpm_remap_chunk_elems(void) { return REMAP_CHUNK; }

extern "C" int ddpm_remap(const ddpm_remap_entry* entries, const int32_t* chunks, int n_chunks, int to_a, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(entries && chunks && n_chunks > 0, "remap: bad arguments");
    hipLaunchKernelGGL(remap_kernel, dim3((unsigned)n_chunks), dim3(256), 0, stream, entries, (const int2*)chunks, to_a);
    return ddpm_check_launch("remap");
}

// Adam with bias correction (torch.optim.Adam semantics, weight_decay = L2 added to the gradient), plus an optional
// exponential moving average ema = lerp(ema, p, 1 - ema_decay) of the updated parameters
// (torch.optim.swa_utils.AveragedModel with get_ema_multi_avg_fn: its first update copies the parameters).  step: device int64, incremented here by the first thread.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ ema, long long n,
                                                    const float* __restrict__ lr_ptr, float lr_host, float b1, float b2, float eps,
                                                    float wd, float ema_decay, float grad_scale, long long* __restrict__ step) {
    const long long st = *step + 1;
    const float lr = lr_ptr ? *lr_ptr : lr_host;
    const float c1 = 1.0f - powf(b1, (float)st), c2 = 1.0f - powf(b2, (float)st);
    const float step_size = lr / c1, inv_sqrt_c2 = rsqrtf(c2);
    const long long n4 = n / 4;
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        f32x4 pp = *(f32x4*)(p + q * 4), gg = *(const f32x4*)(g + q * 4), mm = *(f32x4*)(m + q * 4), vv = *(f32x4*)(v + q * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * grad_scale + wd * pp[j];
            mm[j] = b1 * mm[j] + (1.0f - b1) * gr;
            vv[j] = b2 * vv[j] + (1.0f - b2) * gr * gr;
            pp[j] -= step_size * mm[j] / (sqrtf(vv[j]) * inv_sqrt_c2 + eps);
        }
        *(f32x4*)(p + q * 4) = pp; *(f32x4*)(m + q * 4) = mm; *(f32x4*)(v + q * 4) = vv;
        if (ema) {
            f32x4 ee = *(f32x4*)(ema + q * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) ee[j] = st == 1 ? pp[j] : ee[j] + (1.0f - ema_decay) * (pp[j] - ee[j]);
            *(f32x4*)(ema + q * 4) = ee;
        }
    }
}

__global__ void bump_step_kernel(long long* step) { *step += 1; }

extern "C" int ddpm_adam_step_range(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* lr_dev, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale,
                                    int64_t* step, int bump, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(p && g && m && v && step && n > 0 && n % 4 == 0, "adam: bad arguments (n must be a multiple of 4)");
    long long gx = (n / 4 + 255) / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx), dim3(256), 0, stream, p, g, m, v, ema, (long long)n, lr_dev, lr, beta1, beta2, eps,
                       weight_decay, ema_decay, grad_scale, (long long*)step);
    if (bump) hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, stream, (long long*)step);
    return ddpm_check_launch("adam");
}

extern "C" int ddpm_adam_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* lr_dev, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale,
                              int64_t* step, ddpm_stream_t stream) {
    return ddpm_adam_step_range(p, g, m, v, ema, n, lr_dev, lr, beta1, beta2, eps, weight_decay, ema_decay, grad_scale, step, 1, stream);
}
