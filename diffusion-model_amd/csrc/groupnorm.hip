// GroupNorm fused with activation, dropout mask and residual add, forward and backward, on NHWC bf16.
//
// Work unit = one image x one SLAB of channels (a run of whole groups, up to 64 channels wide so that
// the lanes of a wave read contiguous 16-byte pieces of the same pixel).  A unit is owned by one workgroup
// (256 threads, or a single wave when the unit is small), which makes the statistics deterministic and
// needs no atomics in the forward pass.  Each thread owns one 8-channel vector column of the slab and
// walks the pixels; statistics are fp32, reduced with wave shuffles then across waves through LDS.
// The second pass re-reads the slab (an L2 hit: a slab is at most a few hundred KB) and writes the result.
//
// With groups == 0 the kernels skip normalisation and apply only the activation (the reference's
// conv -> activation sites without a norm: unets.py:50-51,179-180; constituent_blocks.py:309-311).
#include "common.h"
#include "ddpm_hip.h"

#include <mutex>

// Kernel bodies are compiled once per hot activation (A = gelu, silu, none: the activation folds to straight-line code) and
// once with A = -1, where the runtime code selects (every other activation, swish / hardswish with their scalar included).
// A runtime switch inside the unrolled element loops costs a scalar branch per element and keeps the compiler from packing
// the arithmetic: the GroupNorm of a 128x128 image is a stream of a few hundred MB through exactly these loops.
#define GN_ACT(A, a) ((A) >= 0 ? (A) : (a).act)
#define GN_BY_ACT(BODY, ...)                                       \
    do {                                                           \
        if (a.act == ACT_GELU) BODY<__VA_ARGS__, ACT_GELU>(a);     \
        else if (a.act == ACT_SILU) BODY<__VA_ARGS__, ACT_SILU>(a); \
        else if (a.act == ACT_NONE) BODY<__VA_ARGS__, ACT_NONE>(a); \
        else BODY<__VA_ARGS__, -1>(a);                             \
    } while (0)

struct GnArgs {
    const bf16* x;             // [N][HW][C] raw convolution output
    const float* gamma;        // [C]
    const float* beta;         // [C]
    const bf16* res;           // optional residual, added last
    const unsigned char* mask; // optional dropout keep bits of the OUTPUT (1 byte / 8 elements)
    float mask_scale;
    bf16* y;                   // forward output / backward: dx
    float* stats;              // [N][G][2] mean, rstd (written by forward, read by backward)
    // backward only
    const bf16* dy;
    float* part;                   // [N][3][C] fp32 per-image partial sums: dgamma, dbeta, sum of dx (= the producing conv's
                                   // bias gradient); every element is written exactly once, a later pass sums over N
    int N, HW, C, G, SC;           // SC = slab width (channels)
    const float* sp; int NP;       // split forward, optional: [N][G][NP][2] partial sums the producing convolution stored
    int act;
    float eps;
    const float* act_beta;         // the activation's learnable scalar (swish / hardswish), else unused
    float* dbeta_part;             // backward, swish / hardswish: [N][C] per image and channel sums of dy * mask * d act / d scalar
    float* scratch;                // split path: partial sums of the pixel splits
    int P, rows_per_split;         // split path: pixel splits per (image, slab) and pixels per split
    int fixed;                     // statistics are given (stats is an input of the forward pass, a constant of the backward pass)
};

// NV = vectors a thread keeps in registers when the whole slab fits (NV > 0): the slab is then read from memory
// exactly once, with all of a thread's loads in flight together; NV = 0 is the two-pass fallback for huge slabs.
// Small NV with many threads is preferred: these kernels are latency bound, and a workgroup per image leaves a
// CU with a single workgroup, so the waves must come from the workgroup size (up to 512 threads = 8 waves; 1024
// threads would cap the kernel at 128 VGPRs and spill).

template <int THREADS, int NV, int A>
__device__ __forceinline__ void gn_fwd_body(const GnArgs& a) {
    __shared__ float s_red[(THREADS / 64) * 64 < 128 ? 128 : (THREADS / 64) * 64];
    __shared__ float s_mean[32], s_rstd[32];
    const int tid = threadIdx.x;
    const int slabs = a.C / a.SC;
    const int n = blockIdx.x / slabs, c0 = (blockIdx.x - n * slabs) * a.SC;
    const int vps = a.SC / 8;                 // vectors per pixel in the slab (power of two, <= 8)
    const int cv = tid % vps, prow = tid / vps, pstep = THREADS / vps;
    const long long base = (long long)n * a.HW * a.C + c0 + cv * 8;
    const int cpg = a.G > 0 ? a.C / a.G : 1;

    constexpr bool CACHED = NV > 0;
    constexpr int GN_NV = NV > 0 ? NV : 1;
    const float bsc = (A < 0 && ACT_HAS_BETA(a.act)) ? *a.act_beta : 1.f;
    // per-channel affine parameters are fetched up front so their latency overlaps the slab loads
    float gam[8], bet[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        gam[j] = a.G > 0 ? a.gamma[c0 + cv * 8 + j] : 1.f;
        bet[j] = a.G > 0 ? a.beta[c0 + cv * 8 + j] : 0.f;
    }
    bf16x8 xv[GN_NV];
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < GN_NV; ++k) {
            const int p = prow + k * pstep;
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[k][j] = (bf16)0.f;
            if (p < a.HW) xv[k] = *(const bf16x8*)(a.x + base + (long long)p * a.C);
        }
    }
    if (a.G > 0 && a.fixed) {
        const int ngrp = a.SC / cpg;
        if (tid < ngrp) {
            const int g = c0 / cpg + tid;
            s_mean[tid] = a.stats[((long long)n * a.G + g) * 2 + 0];
            s_rstd[tid] = a.stats[((long long)n * a.G + g) * 2 + 1];
        }
        __syncthreads();
    } else if (a.G > 0) {
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
        if constexpr (CACHED) {
#pragma unroll
            for (int k = 0; k < GN_NV; ++k)       // out-of-range slots hold zeros and add nothing
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)xv[k][j];
                    s[j] += f; ss[j] += f * f;
                }
        } else {
#pragma unroll 2
            for (int p = prow; p < a.HW; p += pstep) {
                const bf16x8 v = *(const bf16x8*)(a.x + base + (long long)p * a.C);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)v[j];
                    s[j] += f; ss[j] += f * f;
                }
            }
        }
        block_reduce_cols<THREADS>(s, vps, s_red, tid);
        block_reduce_cols<THREADS>(ss, vps, s_red, tid);
        // per-channel sums of this thread's vector -> LDS, then one thread per group finishes
        __syncthreads();
        if (tid < vps) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s_red[tid * 8 + j] = s[j]; s_red[64 + tid * 8 + j] = ss[j]; }
        }
        __syncthreads();
        const int ngrp = a.SC / cpg;
        if (tid < ngrp) {
            float S = 0.f, SS = 0.f;
            for (int c = 0; c < cpg; ++c) { S += s_red[tid * cpg + c]; SS += s_red[64 + tid * cpg + c]; }
            const float m = 1.0f / (float)(cpg * a.HW);
            const float mean = S * m;
            const float var = fmaxf(SS * m - mean * mean, 0.f);
            const float rstd = rsqrtf(var + a.eps);
            s_mean[tid] = mean; s_rstd[tid] = rstd;
            if (a.stats) {
                const int g = c0 / cpg + tid;
                a.stats[((long long)n * a.G + g) * 2 + 0] = mean;
                a.stats[((long long)n * a.G + g) * 2 + 1] = rstd;
            }
        }
        __syncthreads();
    }
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (a.G > 0) {
            const int cl = cv * 8 + j, g = cl / cpg;
            const float ga = gam[j] * s_rstd[g];
            sc[j] = ga; sh[j] = bet[j] - s_mean[g] * ga;
        } else {
            sc[j] = 1.f; sh[j] = 0.f;
        }
    }
    auto finish = [&](const bf16x8& v, long long e) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = act_fwd((float)v[j] * sc[j] + sh[j], GN_ACT(A, a), bsc);
        if (a.mask) {
            const unsigned m = a.mask[e >> 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (m >> j) & 1 ? o[j] * a.mask_scale : 0.f;
        }
        if (a.res) {
            const bf16x8 r = *(const bf16x8*)(a.res + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += (float)r[j];
        }
        bf16x8 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (bf16)o[j];
        *(bf16x8*)(a.y + e) = w;
    };
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < GN_NV; ++k) {
            const int p = prow + k * pstep;
            if (p < a.HW) finish(xv[k], base + (long long)p * a.C);
        }
    } else {
#pragma unroll 2
        for (int p = prow; p < a.HW; p += pstep) {
            const long long e = base + (long long)p * a.C;
            finish(*(const bf16x8*)(a.x + e), e);
        }
    }
}

template <int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void gn_fwd_kernel(const GnArgs a) { GN_BY_ACT(gn_fwd_body, THREADS, NV); }

// Backward.  With g = dy * mask * act'(z), xhat = (x - mean) rstd, m = cpg*HW:
//   dgamma_c = sum g xhat,  dbeta_c = sum g,
//   dx = rstd * ( gamma_c g - (S1 + xhat S2)/m ),  S1 = sum_group gamma g,  S2 = sum_group gamma g xhat.
template <int THREADS, int NV, int A>
__device__ __forceinline__ void gn_bwd_body(const GnArgs& a) {
    __shared__ float s_red[(THREADS / 64) * 64 < 128 ? 128 : (THREADS / 64) * 64];
    __shared__ float s_s1[32], s_s2[32];
    const int tid = threadIdx.x;
    const int slabs = a.C / a.SC;
    const int n = blockIdx.x / slabs, c0 = (blockIdx.x - n * slabs) * a.SC;
    const int vps = a.SC / 8;
    const int cv = tid % vps, prow = tid / vps, pstep = THREADS / vps;
    const long long base = (long long)n * a.HW * a.C + c0 + cv * 8;
    const int cpg = a.G > 0 ? a.C / a.G : 1;
    // the cached path addresses memory as a workgroup-uniform base (this image, this slab) plus a 32-bit byte offset per lane
    const long long ubase = (long long)n * a.HW * a.C + c0;
    const bf16* const xb = a.x + ubase; const bf16* const dyb = a.dy + ubase; bf16* const yb = a.y + ubase;
    const unsigned char* const mb = a.mask ? a.mask + (ubase >> 3) : nullptr;

    float ga[8], be[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = cv * 8 + j;
        if (a.G > 0) {
            const int g = (c0 + cl) / cpg;
            ga[j] = a.gamma[c0 + cl]; be[j] = a.beta[c0 + cl];
            mean[j] = a.stats[((long long)n * a.G + g) * 2 + 0];
            rstd[j] = a.stats[((long long)n * a.G + g) * 2 + 1];
        } else {
            ga[j] = 1.f; be[j] = 0.f; mean[j] = 0.f; rstd[j] = 1.f;
        }
    }
    const bool hb = A < 0 && ACT_HAS_BETA(a.act);
    const float bsc = hb ? *a.act_beta : 1.f;
    float ab[8];                                  // swish / hardswish: per-channel sums for the scalar's gradient
#pragma unroll
    for (int j = 0; j < 8; ++j) ab[j] = 0.f;
    // g and xhat of one vector; acc: also add this vector's terms of the activation scalar's gradient
    auto gx = [&](const bf16x8& xv, const bf16x8& dv, unsigned m, float (&g)[8], float (&xh)[8], bool acc) {
        const float ms = a.mask ? a.mask_scale : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xh[j] = ((float)xv[j] - mean[j]) * rstd[j];
            const float z = xh[j] * ga[j] + be[j];
            const float gg = (float)dv[j] * act_bwd(z, GN_ACT(A, a), bsc) * ms;
            g[j] = (m >> j) & 1 ? gg : 0.f;
            if constexpr (A < 0)
                if (hb && acc) ab[j] += (m >> j) & 1 ? (float)dv[j] * act_dbeta(z, a.act, bsc) * ms : 0.f;
        }
    };
    // cached path: x stays in registers as bf16, the activation-gradient term g as fp32 (so exp/erf run once)
    constexpr bool CACHED = NV > 0;
    constexpr int GN_NV = NV > 0 ? NV : 1;
    bf16x8 xc[GN_NV];
    float gc[GN_NV][8];
    float S1[8], S2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) S1[j] = S2[j] = 0.f;
    float a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
    if constexpr (CACHED) {
        bf16x8 dc[GN_NV];
        unsigned mc[GN_NV];
#pragma unroll
        for (int k = 0; k < GN_NV; ++k) {
            const int p = prow + k * pstep;
            mc[k] = 0xFFu;                                   // (vectors past the last pixel are never read: every use is guarded)
            if (p < a.HW) {
                const unsigned e = (unsigned)(p * a.C + cv * 8);
                xc[k] = ld_at<bf16x8>(xb, e * 2);
                dc[k] = ld_at<bf16x8>(dyb, e * 2);
                if (mb) mc[k] = ld_at<unsigned char>(mb, e >> 3);
            }
        }
#pragma unroll
        for (int k = 0; k < GN_NV; ++k) {
            const int p = prow + k * pstep;
            float xh[8];
            if (p < a.HW) {
                gx(xc[k], dc[k], mc[k], gc[k], xh, true);
#pragma unroll
                for (int j = 0; j < 8; ++j) { a1[j] += gc[k][j]; a2[j] += gc[k][j] * xh[j]; }
            }
        }
        // x stays packed (bf16) across the reductions below; otherwise its fp32 copies are kept for the second pass as well
#pragma unroll
        for (int k = 0; k < GN_NV; ++k) asm volatile("" : "+v"(xc[k]));
    } else if (a.G > 0) {
#pragma unroll 2
        for (int p = prow; p < a.HW; p += pstep) {
            const long long e = base + (long long)p * a.C;
            float g[8], xh[8];
            gx(*(const bf16x8*)(a.x + e), *(const bf16x8*)(a.dy + e), a.mask ? a.mask[e >> 3] : 0xFFu, g, xh, true);
#pragma unroll
            for (int j = 0; j < 8; ++j) { a1[j] += g[j]; a2[j] += g[j] * xh[j]; }
        }
    }
    if (a.G > 0) {
        block_reduce_cols<THREADS>(a1, vps, s_red, tid);
        block_reduce_cols<THREADS>(a2, vps, s_red, tid);
        __syncthreads();
        if (tid < vps) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cl = tid * 8 + j;
                a.part[((long long)n * 3 + 0) * a.C + c0 + cl] = a2[j];
                a.part[((long long)n * 3 + 1) * a.C + c0 + cl] = a1[j];
                s_red[cl] = a1[j] * ga[j];
                s_red[64 + cl] = a2[j] * ga[j];
            }
        }
        __syncthreads();
        const int ngrp = a.SC / cpg;
        if (tid < ngrp) {
            float u = 0.f, v = 0.f;
            for (int c = 0; c < cpg; ++c) { u += s_red[tid * cpg + c]; v += s_red[64 + tid * cpg + c]; }
            const float im = 1.0f / (float)(cpg * a.HW);
            s_s1[tid] = u * im; s_s2[tid] = v * im;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (cv * 8 + j) / cpg;
            S1[j] = a.fixed ? 0.f : s_s1[g]; S2[j] = a.fixed ? 0.f : s_s2[g];
        }
    }
    float db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) db[j] = 0.f;
    if constexpr (CACHED) {
#pragma unroll
        for (int k = 0; k < GN_NV; ++k) {
            const int p = prow + k * pstep;
            if (p < a.HW) {
                bf16x8 w;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xc[k][j] - mean[j]) * rstd[j];
                    const float dx = rstd[j] * (ga[j] * gc[k][j] - (S1[j] + xh * S2[j]));
                    db[j] += dx;
                    w[j] = (bf16)dx;
                }
                st_at(yb, (unsigned)(p * a.C + cv * 8) * 2, w);
            }
        }
    } else {
#pragma unroll 2
        for (int p = prow; p < a.HW; p += pstep) {
            const long long e = base + (long long)p * a.C;
            float g[8], xh[8];
            gx(*(const bf16x8*)(a.x + e), *(const bf16x8*)(a.dy + e), a.mask ? a.mask[e >> 3] : 0xFFu, g, xh, a.G == 0);   // without a norm this is the only pass
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dx = rstd[j] * (ga[j] * g[j] - (S1[j] + xh[j] * S2[j]));
                db[j] += dx;
                w[j] = (bf16)dx;
            }
            *(bf16x8*)(a.y + e) = w;
        }
    }
    block_reduce_cols<THREADS>(db, vps, s_red, tid);
    if (tid < vps) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a.part[((long long)n * 3 + 2) * a.C + c0 + tid * 8 + j] = db[j];
    }
    if (hb) {
        block_reduce_cols<THREADS>(ab, vps, s_red, tid);
        if (tid < vps) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a.dbeta_part[(long long)n * a.C + c0 + tid * 8 + j] = ab[j];
        }
    }
}

template <int THREADS, int NV>
__global__ __launch_bounds__(THREADS) void gn_bwd_kernel(const GnArgs a) { GN_BY_ACT(gn_bwd_body, THREADS, NV); }

// ---- split path for large images --------------------------------------------------------------------------------
// One workgroup per (image, slab) walks a 64x64 or 128x128 image alone for hundreds of microseconds while most of the
// machine idles.  For such slabs the pixels are cut into P splits: a first launch reduces every split to partial sums
// in a caller-provided scratch buffer, a second launch adds the P partials of its groups / channels in a fixed order
// (deterministic, no atomics) and applies the result to its own split.  Both launches stream at full width.
constexpr int GS_T = 256;
constexpr int GS_U = 4;          // rows per thread whose loads are in flight together

__global__ __launch_bounds__(GS_T) void gn_fwd_split_stats_kernel(const GnArgs a) {
    __shared__ float s_red[(GS_T / 64) * 64 < 128 ? 128 : (GS_T / 64) * 64];
    const int tid = threadIdx.x;
    const int slabs = a.C / a.SC;
    int b = blockIdx.x;
    const int p = b % a.P; b /= a.P;
    const int n = b / slabs, c0 = (b - n * slabs) * a.SC;
    const int vps = a.SC / 8, cv = tid % vps, prow = tid / vps, pstep = GS_T / vps;
    const long long base = (long long)n * a.HW * a.C + c0 + cv * 8;
    const int cpg = a.C / a.G;
    const int r0 = p * a.rows_per_split, r1 = min(a.HW, r0 + a.rows_per_split);
    float s[8], ss[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
#pragma unroll 4
    for (int r = r0 + prow; r < r1; r += pstep) {
        const bf16x8 v = *(const bf16x8*)(a.x + base + (long long)r * a.C);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s[j] += f; ss[j] += f * f; }
    }
    block_reduce_cols<GS_T>(s, vps, s_red, tid);
    block_reduce_cols<GS_T>(ss, vps, s_red, tid);
    __syncthreads();
    if (tid < vps) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s_red[tid * 8 + j] = s[j]; s_red[64 + tid * 8 + j] = ss[j]; }
    }
    __syncthreads();
    const int ngrp = a.SC / cpg;
    if (tid < ngrp) {
        float S = 0.f, SS = 0.f;
        for (int c = 0; c < cpg; ++c) { S += s_red[tid * cpg + c]; SS += s_red[64 + tid * cpg + c]; }
        const int g = c0 / cpg + tid;
        float* o = a.scratch + (((long long)n * a.G + g) * a.P + p) * 2;
        o[0] = S; o[1] = SS;
    }
}

template <int T_, int A>
__device__ __forceinline__ void gn_fwd_split_apply_body(const GnArgs& a) {
    __shared__ float s_mean[32], s_rstd[32];
    const int tid = threadIdx.x;
    const int slabs = a.C / a.SC;
    int b = blockIdx.x;
    const int p = b % a.P; b /= a.P;
    const int n = b / slabs, c0 = (b - n * slabs) * a.SC;
    const int vps = a.SC / 8, cv = tid % vps, prow = tid / vps, pstep = GS_T / vps;
    const long long base = (long long)n * a.HW * a.C + c0 + cv * 8;
    const int cpg = a.G > 0 ? a.C / a.G : 1;
    if (a.G > 0) {
        const int ngrp = a.SC / cpg;
        if (a.sp) {
            // the producing convolution left NP partial pairs per (image, group).  A slab holds 1, 2, 4, ... 32 groups: the
            // workgroup's threads share them out, and each group's threads fold their sums in a fixed order (deterministic)
            __shared__ float s_fs[32], s_fss[32];
            const int tpg = GS_T / ngrp;                                  // 8 .. 256 threads per group
            const int gi = tid / tpg, k = tid - gi * tpg;
            const float2* in = (const float2*)a.sp + ((long long)n * a.G + c0 / cpg + gi) * a.NP;
            float S = 0.f, SS = 0.f;
            for (int q = k; q < a.NP; q += tpg) { const float2 v = in[q]; S += v.x; SS += v.y; }
            for (int o = 1; o < 64 && o < tpg; o <<= 1) { S += __shfl_xor(S, o); SS += __shfl_xor(SS, o); }
            const int lpg = tpg < 64 ? tpg : 64;                          // lanes behind each of the sums that are left
            if (tid % lpg == 0) { s_fs[tid / lpg] = S; s_fss[tid / lpg] = SS; }
            __syncthreads();
            if (tid < ngrp) {
                const int per = tpg / lpg, g = c0 / cpg + tid;
                S = SS = 0.f;
                for (int i = 0; i < per; ++i) { S += s_fs[tid * per + i]; SS += s_fss[tid * per + i]; }
                const float m = 1.0f / (float)(cpg * a.HW);
                const float mean = S * m;
                const float rstd = rsqrtf(fmaxf(SS * m - mean * mean, 0.f) + a.eps);
                s_mean[tid] = mean; s_rstd[tid] = rstd;
                if (p == 0 && a.stats) {
                    a.stats[((long long)n * a.G + g) * 2 + 0] = mean;
                    a.stats[((long long)n * a.G + g) * 2 + 1] = rstd;
                }
            }
        } else if (tid < ngrp && a.fixed) {
            const int g = c0 / cpg + tid;
            s_mean[tid] = a.stats[((long long)n * a.G + g) * 2 + 0];
            s_rstd[tid] = a.stats[((long long)n * a.G + g) * 2 + 1];
        } else if (tid < ngrp) {
            const int g = c0 / cpg + tid;
            const float* in = a.scratch + ((long long)n * a.G + g) * a.P * 2;
            float S = 0.f, SS = 0.f;
            for (int q = 0; q < a.P; ++q) { S += in[q * 2]; SS += in[q * 2 + 1]; }
            const float m = 1.0f / (float)(cpg * a.HW);
            const float mean = S * m;
            const float rstd = rsqrtf(fmaxf(SS * m - mean * mean, 0.f) + a.eps);
            s_mean[tid] = mean; s_rstd[tid] = rstd;
            if (p == 0 && a.stats) {
                a.stats[((long long)n * a.G + g) * 2 + 0] = mean;
                a.stats[((long long)n * a.G + g) * 2 + 1] = rstd;
            }
        }
        __syncthreads();
    }
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (a.G > 0) {
            const int cl = cv * 8 + j, g = cl / cpg;
            const float ga = a.gamma[c0 + cl] * s_rstd[g];
            sc[j] = ga; sh[j] = a.beta[c0 + cl] - s_mean[g] * ga;
        } else {
            sc[j] = 1.f; sh[j] = 0.f;
        }
    }
    const int r0 = p * a.rows_per_split, r1 = min(a.HW, r0 + a.rows_per_split);
    const float bsc = (A < 0 && ACT_HAS_BETA(a.act)) ? *a.act_beta : 1.f;
    // A workgroup's strip is a few rows per thread: all of a batch's loads are issued before the first activation is evaluated
    // (one round trip to memory per batch instead of one per row; rows past the strip's end re-read its last row and are skipped)
    for (int rb = r0 + prow; rb < r1; rb += GS_U * pstep) {
        bf16x8 v[GS_U], rv[GS_U];
        unsigned m[GS_U];
#pragma unroll
        for (int u = 0; u < GS_U; ++u) {
            const long long e = base + (long long)min(rb + u * pstep, r1 - 1) * a.C;
            v[u] = *(const bf16x8*)(a.x + e);
            m[u] = a.mask ? a.mask[e >> 3] : 0xFFu;
            if (a.res) rv[u] = *(const bf16x8*)(a.res + e);
        }
#pragma unroll
        for (int u = 0; u < GS_U; ++u) {
            const int r = rb + u * pstep;
            if (r >= r1) break;
            const long long e = base + (long long)r * a.C;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = act_fwd((float)v[u][j] * sc[j] + sh[j], GN_ACT(A, a), bsc);
            if (a.mask) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (m[u] >> j) & 1 ? o[j] * a.mask_scale : 0.f;
            }
            if (a.res) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += (float)rv[u][j];
            }
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (bf16)o[j];
            *(bf16x8*)(a.y + e) = w;
        }
    }
}

__global__ __launch_bounds__(GS_T) void gn_fwd_split_apply_kernel(const GnArgs a) { GN_BY_ACT(gn_fwd_split_apply_body, GS_T); }

// g = dy * mask * act'(z) and xhat of one vector (shared by the two backward split kernels)
struct GnBwdLane {
    float ga[8], be[8], mean[8], rstd[8];
    float bsc;                                    // the activation's scalar (1 for the parameter-free ones)
};
__device__ __forceinline__ void gn_bwd_lane_init(const GnArgs& a, int n, int c, GnBwdLane& L) {
    const int cpg = a.G > 0 ? a.C / a.G : 1;
    L.bsc = ACT_HAS_BETA(a.act) ? *a.act_beta : 1.f;       // (the specialised bodies never read it)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (a.G > 0) {
            const int g = (c + j) / cpg;
            L.ga[j] = a.gamma[c + j]; L.be[j] = a.beta[c + j];
            L.mean[j] = a.stats[((long long)n * a.G + g) * 2 + 0];
            L.rstd[j] = a.stats[((long long)n * a.G + g) * 2 + 1];
        } else {
            L.ga[j] = 1.f; L.be[j] = 0.f; L.mean[j] = 0.f; L.rstd[j] = 1.f;
        }
    }
}
// with_ab: also adds this vector's terms of the activation scalar's gradient to ab (an array by reference: a pointer that
// may be null would put the caller's array in scratch memory)
struct GnBwdIn { bf16x8 xv, dv; unsigned m; };
__device__ __forceinline__ GnBwdIn gn_bwd_load(const GnArgs& a, long long e) {
    GnBwdIn q;
    q.xv = *(const bf16x8*)(a.x + e); q.dv = *(const bf16x8*)(a.dy + e);
    q.m = a.mask ? a.mask[e >> 3] : 0xFFu;
    return q;
}
template <int A>
__device__ __forceinline__ void gn_bwd_gx(const GnArgs& a, const GnBwdLane& L, const GnBwdIn& q, float (&g)[8], float (&xh)[8], float (&xf)[8],
                                          float (&ab)[8], bool with_ab) {
    const float ms = a.mask ? a.mask_scale : 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xf[j] = (float)q.xv[j];
        xh[j] = (xf[j] - L.mean[j]) * L.rstd[j];
        const float z = xh[j] * L.ga[j] + L.be[j];
        const float gg = (float)q.dv[j] * act_bwd(z, GN_ACT(A, a), L.bsc) * ms;
        g[j] = (q.m >> j) & 1 ? gg : 0.f;
        if constexpr (A < 0)
            if (with_ab) ab[j] += (q.m >> j) & 1 ? (float)q.dv[j] * act_dbeta(z, a.act, L.bsc) * ms : 0.f;
    }
}

// rows of the backward scratch per (image, split): sum g, sum g*xhat, sum x and, for swish / hardswish, the scalar-gradient terms
__device__ __host__ __forceinline__ int gn_bwd_rows(int act) { return ACT_HAS_BETA(act) ? 4 : 3; }

// scratch layout (backward): [N][P][rows][C] per split and channel
template <int T_, int A>
__device__ __forceinline__ void gn_bwd_split_sums_body(const GnArgs& a) {
    __shared__ float s_red[(GS_T / 64) * 64 < 128 ? 128 : (GS_T / 64) * 64];
    const int tid = threadIdx.x;
    const int slabs = a.C / a.SC;
    int b = blockIdx.x;
    const int p = b % a.P; b /= a.P;
    const int n = b / slabs, c0 = (b - n * slabs) * a.SC;
    const int vps = a.SC / 8, cv = tid % vps, prow = tid / vps, pstep = GS_T / vps;
    const long long base = (long long)n * a.HW * a.C + c0 + cv * 8;
    GnBwdLane L;
    gn_bwd_lane_init(a, n, c0 + cv * 8, L);
    float a1[8], a2[8], sx[8], ab[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = sx[j] = ab[j] = 0.f;
    const bool hb = A < 0 && ACT_HAS_BETA(a.act);
    const int R = gn_bwd_rows(a.act);
    const int r0 = p * a.rows_per_split, r1 = min(a.HW, r0 + a.rows_per_split);
    // g = dy * mask * act'(z) is evaluated once per element, here: it goes to the result tensor in bf16 and the second launch
    // turns it into the input gradient in place (the activation's derivative is most of either launch's arithmetic).
    // A batch's loads are all in flight before its first row is evaluated; rows past the strip's end re-read its last row and
    // are skipped.
    constexpr int UB = 2;
    for (int rb = r0 + prow; rb < r1; rb += UB * pstep) {
        GnBwdIn in[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) in[u] = gn_bwd_load(a, base + (long long)min(rb + u * pstep, r1 - 1) * a.C);
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = rb + u * pstep;
            if (r >= r1) break;
            float g[8], xh[8], xf[8];
            gn_bwd_gx<A>(a, L, in[u], g, xh, xf, ab, hb);
            bf16x8 gv;
#pragma unroll
            for (int j = 0; j < 8; ++j) { gv[j] = (bf16)g[j]; a1[j] += g[j]; a2[j] += g[j] * xh[j]; sx[j] += xf[j]; }
            *(bf16x8*)(a.y + base + (long long)r * a.C) = gv;
        }
    }
    block_reduce_cols<GS_T>(a1, vps, s_red, tid);
    block_reduce_cols<GS_T>(a2, vps, s_red, tid);
    block_reduce_cols<GS_T>(sx, vps, s_red, tid);
    if (hb) block_reduce_cols<GS_T>(ab, vps, s_red, tid);
    if (tid < vps) {
        float* o = a.scratch + (((long long)n * a.P + p) * R) * a.C + c0 + tid * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { o[j] = a1[j]; o[a.C + j] = a2[j]; o[2 * a.C + j] = sx[j]; }
        if (hb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[3 * a.C + j] = ab[j];
        }
    }
}

__global__ __launch_bounds__(GS_T) void gn_bwd_split_sums_kernel(const GnArgs a) { GN_BY_ACT(gn_bwd_split_sums_body, GS_T); }

template <int T_, int A>
__device__ __forceinline__ void gn_bwd_split_apply_body(const GnArgs& a) {
    __shared__ float s_a1[64], s_a2[64], s_sx[64], s_s1[32], s_s2[32];
    const int tid = threadIdx.x;
    const int slabs = a.C / a.SC;
    int b = blockIdx.x;
    const int p = b % a.P; b /= a.P;
    const int n = b / slabs, c0 = (b - n * slabs) * a.SC;
    const int vps = a.SC / 8, cv = tid % vps, prow = tid / vps, pstep = GS_T / vps;
    const long long base = (long long)n * a.HW * a.C + c0 + cv * 8;
    const int cpg = a.G > 0 ? a.C / a.G : 1;
    if (tid < a.SC) {                                 // per-channel totals: the P partials in a fixed order
        const int R = gn_bwd_rows(a.act);
        const float* in = a.scratch + ((long long)n * a.P * R) * a.C + c0 + tid;
        float u = 0.f, v = 0.f, w = 0.f;
        for (int q = 0; q < a.P; ++q) { u += in[(q * R + 0) * a.C]; v += in[(q * R + 1) * a.C]; w += in[(q * R + 2) * a.C]; }
        s_a1[tid] = u; s_a2[tid] = v; s_sx[tid] = w;
        if (R == 4 && p == 0) {
            float t = 0.f;
            for (int q = 0; q < a.P; ++q) t += in[(q * R + 3) * a.C];
            a.dbeta_part[(long long)n * a.C + c0 + tid] = t;
        }
    }
    __syncthreads();
    if (a.G > 0) {
        const int ngrp = a.SC / cpg;
        if (tid < ngrp) {
            float u = 0.f, v = 0.f;
            for (int c = 0; c < cpg; ++c) {
                const float ga = a.gamma[c0 + tid * cpg + c];
                u += s_a1[tid * cpg + c] * ga; v += s_a2[tid * cpg + c] * ga;
            }
            const float im = a.fixed ? 0.f : 1.0f / (float)(cpg * a.HW);
            s_s1[tid] = u * im; s_s2[tid] = v * im;
        }
        __syncthreads();
    }
    if (p == 0 && tid < a.SC) {                       // per-image parameter-gradient partial sums
        const int c = c0 + tid;
        float dbias = s_a1[tid];
        if (a.G > 0) {
            const int g = c / cpg, gl = tid / cpg;
            const float mean = a.stats[((long long)n * a.G + g) * 2], rstd = a.stats[((long long)n * a.G + g) * 2 + 1];
            a.part[((long long)n * 3 + 0) * a.C + c] = s_a2[tid];
            a.part[((long long)n * 3 + 1) * a.C + c] = s_a1[tid];
            // sum over pixels of dx_c = rstd (gamma_c g - S1 - xhat S2), with sum xhat_c = rstd (sum x_c - HW mean)
            dbias = rstd * (a.gamma[c] * s_a1[tid] - (float)a.HW * s_s1[gl] - s_s2[gl] * rstd * (s_sx[tid] - (float)a.HW * mean));
        }
        a.part[((long long)n * 3 + 2) * a.C + c] = dbias;
    }
    // dx = rstd (gamma g - S1 - xhat S2) from x and the g the first launch left in the result tensor, in place; per channel:
    // dx = k1 g - (k0 + k2 x) with k1 = rstd gamma, k2 = rstd^2 S2, k0 = rstd (S1 - mean rstd S2)
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cl = cv * 8 + j, g = cl / cpg;
        float mean = 0.f, rstd = 1.f, ga = 1.f, s1 = 0.f, s2 = 0.f;
        if (a.G > 0) {
            mean = a.stats[((long long)n * a.G + c0 / cpg + g) * 2]; rstd = a.stats[((long long)n * a.G + c0 / cpg + g) * 2 + 1];
            ga = a.gamma[c0 + cl]; s1 = s_s1[g]; s2 = s_s2[g];
        }
        k1[j] = rstd * ga; k2[j] = rstd * rstd * s2; k0[j] = rstd * (s1 - mean * rstd * s2);
    }
    const int r0 = p * a.rows_per_split, r1 = min(a.HW, r0 + a.rows_per_split);
    for (int rb = r0 + prow; rb < r1; rb += GS_U * pstep) {
        bf16x8 xv[GS_U], gv[GS_U];
#pragma unroll
        for (int u = 0; u < GS_U; ++u) {
            const long long e = base + (long long)min(rb + u * pstep, r1 - 1) * a.C;
            xv[u] = *(const bf16x8*)(a.x + e); gv[u] = *(const bf16x8*)(a.y + e);
        }
#pragma unroll
        for (int u = 0; u < GS_U; ++u) {
            const int r = rb + u * pstep;
            if (r >= r1) break;
            bf16x8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (bf16)(k1[j] * (float)gv[u][j] - (k0[j] + k2[j] * (float)xv[u][j]));
            *(bf16x8*)(a.y + base + (long long)r * a.C) = w;
        }
    }
}

__global__ __launch_bounds__(GS_T) void gn_bwd_split_apply_kernel(const GnArgs a) { gn_bwd_split_apply_body<GS_T, 0>(a); }

// splits of a slab's pixels on the split path (0 = the slab is small enough for one workgroup)
static int gn_splits(const GnArgs& a) {
    const long long vecs = (long long)a.HW * (a.SC / 8);
    if (vecs <= 4096) return 0;
    long long p = (vecs + 2047) / 2048;
    return (int)(p > 64 ? 64 : p);
}

static int gn_prepare(GnArgs& a, const ddpm_gn_desc* d, const char* what) {
    DDPM_REQUIRE(d && d->x && d->y, "%s: null pointer", what);
    DDPM_REQUIRE(d->N > 0 && d->HW > 0 && d->C > 0 && d->C % 8 == 0, "%s: C (%d) must be a positive multiple of 8", what, d->C);
    DDPM_REQUIRE(d->G >= 0 && (d->G == 0 || (d->C % d->G == 0 && (d->gamma && d->beta && d->stats))),
                 "%s: groups %d must divide C %d (and gamma/beta/stats must be given)", what, d->G, d->C);
    // slab width: the widest of 64/32/16/8 channels that holds whole groups and whose slab (HW x SC) fits the
    // register cache of a 512-thread workgroup; if none fits, the narrowest valid width on the two-pass path
    const int cpg = d->G > 0 ? d->C / d->G : 1;
    if (d->G > 0) DDPM_REQUIRE(cpg <= 64 && (cpg >= 8 ? cpg % 8 == 0 : 8 % cpg == 0), "%s: unsupported channels per group %d", what, cpg);
    int sc = 0, sc_min = 0;
    for (int c = 64; c >= 8; c >>= 1) {
        if (d->C % c || c % cpg || c / cpg > 32) continue;
        sc_min = c;
        if (!sc && (long long)d->HW * (c / 8) <= 2048) sc = c;
    }
    if (!sc) sc = sc_min;
    // a slab narrower than 64 bytes per pixel uses a fraction of every 128-byte line (and its sibling slabs fetch the
    // same lines again): for big images prefer the widest slab on the two-pass path, whose second read is an L2 hit
    {
        int widest = 0;
        for (int c = 64; c >= 8; c >>= 1)
            if (!(d->C % c) && !(c % cpg) && c / cpg <= 32) { widest = c; break; }
        if (sc < 32 && widest > sc) sc = widest;
    }
    DDPM_REQUIRE(sc > 0, "%s: no slab width for C=%d, groups=%d", what, d->C, d->G);
    a.x = (const bf16*)d->x; a.gamma = d->gamma; a.beta = d->beta; a.res = (const bf16*)d->res; a.mask = d->mask;
    a.mask_scale = d->mask_scale; a.y = (bf16*)d->y; a.stats = d->stats; a.dy = (const bf16*)d->dy; a.part = d->part;
    a.N = d->N; a.HW = d->HW; a.C = d->C; a.G = d->G; a.SC = sc; a.act = d->act;
    a.eps = d->eps;
    DDPM_REQUIRE(d->act >= 0 && d->act <= ACT_LAST, "%s: unknown activation %d", what, d->act);
    DDPM_REQUIRE(!ACT_HAS_BETA(d->act) || d->act_beta, "%s: swish / hardswish need their scalar (act_beta)", what);
    a.act_beta = d->act_beta; a.dbeta_part = d->dbeta_part;
    a.scratch = d->scratch; a.P = 0; a.rows_per_split = 0;
    a.sp = nullptr; a.NP = 0;
    a.fixed = d->fixed_stats != 0;
    DDPM_REQUIRE(!a.fixed || (d->G > 0 && d->stats), "%s: fixed_stats needs groups and the stats tensor", what);
    return DDPM_OK;
}

// floats of scratch the split path needs for this problem (0: the single-workgroup path is used)
static long long gn_scratch_need(const GnArgs& a, bool backward) {
    const int P = gn_splits(a);
    if (!P) return 0;
    return backward ? (long long)a.N * P * gn_bwd_rows(a.act) * a.C : (long long)a.N * (a.G > 0 ? a.G : 0) * P * 2;
}

extern "C" int64_t ddpm_gn_scratch_floats(const ddpm_gn_desc* d, int backward) {
    GnArgs a;
    ddpm_gn_desc t = *d;
    if (!t.x) t.x = (const void*)16;                  // sizing query: pointers are not dereferenced
    if (!t.y) t.y = (void*)16;
    if (t.G > 0) { if (!t.gamma) t.gamma = (const float*)16; if (!t.beta) t.beta = (const float*)16; if (!t.stats) t.stats = (float*)16; }
    if (ACT_HAS_BETA(t.act) && !t.act_beta) t.act_beta = (const float*)16;
    if (gn_prepare(a, &t, "gn_scratch_floats")) return -1;
    const long long need = gn_scratch_need(a, backward != 0);
    return need > 0 ? need : (gn_splits(a) ? 1 : 0);   // activation-only forward needs no partial sums but still takes the split path
}

// true when the split path runs (enough scratch was given)
static bool gn_use_split(GnArgs& a, const ddpm_gn_desc* d, bool backward) {
    const int P = gn_splits(a);
    if (!P) return false;
    const long long need = gn_scratch_need(a, backward);
    if (need > 0 && (!d->scratch || d->scratch_floats < need)) return false;
    a.P = P;
    a.rows_per_split = (a.HW + P - 1) / P;
    return true;
}

// Workgroup size / vectors per thread: aim for 2 vectors per thread (more waves, still two loads in flight each);
// slabs beyond 1024 x 8 vectors take the two-pass path.
#define GN_DISPATCH(KERNEL)                                                                                          \
    const int units = a.N * (a.C / a.SC);                                                                            \
    const long long vecs = (long long)a.HW * (a.SC / 8);                                                             \
    if (vecs <= 64) hipLaunchKernelGGL((KERNEL<64, 1>), dim3(units), dim3(64), 0, stream, a);                        \
    else if (vecs <= 128) hipLaunchKernelGGL((KERNEL<64, 2>), dim3(units), dim3(64), 0, stream, a);                  \
    else if (vecs <= 512) hipLaunchKernelGGL((KERNEL<256, 2>), dim3(units), dim3(256), 0, stream, a);                \
    else if (vecs <= 1024) hipLaunchKernelGGL((KERNEL<512, 2>), dim3(units), dim3(512), 0, stream, a);               \
    else if (vecs <= 2048) hipLaunchKernelGGL((KERNEL<512, 4>), dim3(units), dim3(512), 0, stream, a);               \
    else if (vecs <= 4096) hipLaunchKernelGGL((KERNEL<512, 8>), dim3(units), dim3(512), 0, stream, a);               \
    else hipLaunchKernelGGL((KERNEL<512, 0>), dim3(units), dim3(512), 0, stream, a);

extern "C" int ddpm_gn_act_forward(const ddpm_gn_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GnArgs a;
    if (int rc = gn_prepare(a, d, "gn_forward")) return rc;
    if (gn_use_split(a, d, false)) {
        const int units = a.N * (a.C / a.SC) * a.P;
        if (d->stat_part && d->stat_parts > 0 && a.G > 0 && !a.fixed && GS_T % (a.SC / (a.C / a.G)) == 0) {
            a.sp = d->stat_part; a.NP = d->stat_parts;        // the producing convolution has summed already
        } else if (a.G > 0 && !a.fixed) {
            hipLaunchKernelGGL(gn_fwd_split_stats_kernel, dim3(units), dim3(GS_T), 0, stream, a);
        }
        hipLaunchKernelGGL(gn_fwd_split_apply_kernel, dim3(units), dim3(GS_T), 0, stream, a);
        return ddpm_check_launch("gn_forward");
    }
    GN_DISPATCH(gn_fwd_kernel)
    return ddpm_check_launch("gn_forward");
}

// One wave per (image, group): the lanes share out the producer's partial pairs, add them up in a fixed order and write the
// group's statistics and the (scale, shift) pair of each of its channels.
__global__ __launch_bounds__(64) void gn_fold_affine_kernel(const float* __restrict__ sp, int NP, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int HW, int C, int G, float eps,
                                                            float* __restrict__ stats, float* __restrict__ affine) {
    const int n = blockIdx.x / G, g = blockIdx.x - n * G, lane = threadIdx.x;
    const int cpg = C / G;
    const float2* in = (const float2*)sp + (long long)blockIdx.x * NP;
    float S = 0.f, SS = 0.f;
    for (int q = lane; q < NP; q += 64) { const float2 v = in[q]; S += v.x; SS += v.y; }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { S += __shfl_xor(S, o); SS += __shfl_xor(SS, o); }
    const float m = 1.0f / (float)(cpg * HW);
    const float mean = S * m;
    const float rstd = rsqrtf(fmaxf(SS * m - mean * mean, 0.f) + eps);
    if (lane == 0 && stats) { stats[(long long)blockIdx.x * 2] = mean; stats[(long long)blockIdx.x * 2 + 1] = rstd; }
    for (int c = lane; c < cpg; c += 64) {
        const int ch = g * cpg + c;
        const float ga = gamma[ch] * rstd;
        float* o = affine + ((long long)n * C + ch) * 2;
        o[0] = ga; o[1] = beta[ch] - mean * ga;
    }
}

extern "C" int ddpm_gn_fold_affine(const float* stat_part, int parts, const float* gamma, const float* beta, int N, int HW, int C, int G,
                                   float eps, float* stats, float* affine, ddpm_stream_t stream_) {
    DDPM_REQUIRE(stat_part && gamma && beta && affine, "gn_fold_affine: null pointer");
    DDPM_REQUIRE(parts > 0 && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "gn_fold_affine: bad shape (parts %d, N %d, HW %d, C %d, G %d)", parts, N, HW, C, G);
    hipLaunchKernelGGL(gn_fold_affine_kernel, dim3((unsigned)(N * G)), dim3(64), 0, (hipStream_t)stream_, stat_part, parts, gamma, beta, HW, C, G, eps, stats, affine);
    return ddpm_check_launch("gn_fold_affine");
}

extern "C" int ddpm_gn_act_backward(const ddpm_gn_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GnArgs a;
    if (int rc = gn_prepare(a, d, "gn_backward")) return rc;
    DDPM_REQUIRE(d->dy, "gn_backward: null dy");
    DDPM_REQUIRE(a.part, "gn_backward: partial-sum workspace required");
    DDPM_REQUIRE((const void*)d->y != d->x, "gn_backward: the result may overwrite dy, not x (large images keep an intermediate in it while x is still read)");
    DDPM_REQUIRE(!ACT_HAS_BETA(a.act) || a.dbeta_part, "gn_backward: swish / hardswish need the [N][C] workspace of the scalar's gradient (dbeta_part)");
    if (gn_use_split(a, d, true)) {
        const int units = a.N * (a.C / a.SC) * a.P;
        hipLaunchKernelGGL(gn_bwd_split_sums_kernel, dim3(units), dim3(GS_T), 0, stream, a);
        hipLaunchKernelGGL(gn_bwd_split_apply_kernel, dim3(units), dim3(GS_T), 0, stream, a);
        return ddpm_check_launch("gn_backward");
    }
    GN_DISPATCH(gn_bwd_kernel)
    return ddpm_check_launch("gn_backward");
}

// ------------------------------------------------------------------------------------------------------------------
// The forward pass for callers that cannot bring (or allocate) a workspace: the split path's partial sums live in a
// file-scope device array instead of ddpm_gn_desc.scratch.  Every GPU the library runs on holds its own copy of the array
// (at its own address) until the process ends.  The array's name is only a host-side handle: its device address is asked
// for once per device, with that device current, and kept.
#define GN_STATIC_FLOATS (1 << 20)                                   /* 4 MiB per GPU */
static __device__ __align__(256) float gn_static_scratch[GN_STATIC_FLOATS];

#define GN_MAX_DEVICES 64
static float* g_static_ptr[GN_MAX_DEVICES];                          // device address of gn_static_scratch, per device
static std::mutex g_static_mu;

static int gn_static_pointer(float** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GN_MAX_DEVICES) {
        ddpm_set_error("gn_forward_static: cannot tell the current device (%d)", dev);
        return DDPM_EHIP;
    }
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (!g_static_ptr[dev]) {
        float* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(gn_static_scratch));
        if (e != hipSuccess || !p) {
            ddpm_set_error("gn_forward_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return DDPM_EHIP;
        }
        g_static_ptr[dev] = p;
    }
    *out = g_static_ptr[dev];
    return DDPM_OK;
}

extern "C" int64_t ddpm_gn_static_scratch_floats(void) { return GN_STATIC_FLOATS; }

extern "C" int ddpm_gn_act_forward_static(const ddpm_gn_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GnArgs a;
    if (int rc = gn_prepare(a, d, "gn_forward_static")) return rc;
    const long long need = gn_scratch_need(a, false);
    DDPM_REQUIRE(need <= GN_STATIC_FLOATS,
                 "gn_forward_static: this shape needs %lld floats of partial sums, the built-in workspace holds %d "
                 "(N * groups * pixel splits * 2; use ddpm_gn_act_forward with a workspace of your own)", need, GN_STATIC_FLOATS);
    float* ws = nullptr;
    if (int rc = gn_static_pointer(&ws)) return rc;
    ddpm_gn_desc t = *d;
    t.scratch = ws;                        // the device address, never the array's name
    t.scratch_floats = GN_STATIC_FLOATS;
    a.scratch = ws;
    if (gn_use_split(a, &t, false)) {
        // every partial-sum slot the apply launch reads is written by the statistics launch before it: nothing in the
        // workspace has to start at zero, whatever an earlier (larger) call left there
        const int units = a.N * (a.C / a.SC) * a.P;
        if (a.G > 0 && !a.fixed) hipLaunchKernelGGL(gn_fwd_split_stats_kernel, dim3(units), dim3(GS_T), 0, stream, a);
        hipLaunchKernelGGL(gn_fwd_split_apply_kernel, dim3(units), dim3(GS_T), 0, stream, a);
        return ddpm_check_launch("gn_forward_static");
    }
    GN_DISPATCH(gn_fwd_kernel)
    return ddpm_check_launch("gn_forward_static");
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d in training mode is the GroupNorm above on the tensor viewed as ONE image of N*H*W pixels with one group per
// channel (statistics per channel over batch and space; torch.nn.BatchNorm2d as used by the autoencoder blocks,
// neural_nets/autoencoders/constituent_blocks.py:67,85,160,178 and base_ae.py:192 of the reference).  What remains is the
// module's running-statistics update: running = (1 - momentum) running + momentum * batch value, the variance unbiased
// (count / (count - 1)), and the batch counter.
__global__ __launch_bounds__(256) void bn_update_running_kernel(const float* __restrict__ stats, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, long long* __restrict__ num_batches,
                                                                 int C, float count, float momentum, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        const float mean = stats[c * 2], rstd = stats[c * 2 + 1];
        const float var = fmaxf(1.0f / (rstd * rstd) - eps, 0.f);
        const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (c == 0 && num_batches) *num_batches += 1;
}

extern "C" int ddpm_bn_update_running(const float* stats, float* running_mean, float* running_var, int64_t* num_batches, int C,
                                      int64_t count, float momentum, float eps, ddpm_stream_t stream_) {
    DDPM_REQUIRE(stats && running_mean && running_var && C > 0 && count > 0, "bn_update_running: bad arguments");
    hipLaunchKernelGGL(bn_update_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, stats, running_mean, running_var,
                       (long long*)num_batches, C, (float)count, momentum, eps);
    return ddpm_check_launch("bn_update_running");
}

// stats[c] = (running_mean[c], rsqrt(running_var[c] + eps)): what the forward pass takes as fixed statistics in eval mode
__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                             float* __restrict__ stats, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        stats[c * 2] = running_mean[c];
        stats[c * 2 + 1] = rsqrtf(running_var[c] + eps);
    }
}

extern "C" int ddpm_bn_eval_stats(const float* running_mean, const float* running_var, float* stats, int C, float eps, ddpm_stream_t stream_) {
    DDPM_REQUIRE(stats && running_mean && running_var && C > 0, "bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream_, running_mean, running_var, stats, C, eps);
    return ddpm_check_launch("bn_eval_stats");
}
