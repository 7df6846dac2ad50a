// Depthwise 3x3 convolution (stride 1, padding 1) on NHWC bf16: an HBM-bound 9-tap stencil, not a GEMM.
// One thread produces 8 consecutive channels of one pixel from nine 16-byte neighbour loads (the 3x reuse
// along each axis is served by L1/L2; the tensor is read from HBM once).  The data-gradient is the same
// kernel with tap-flipped weights.  The weight-gradient is a per-channel 9-tap reduction over all pixels:
// each thread keeps 72 + 8 partial sums for its 8 channels over a strip of pixels, lanes that share a
// channel vector are combined with wave shuffles, and one atomic per (tap, channel) per wave lands in HBM.
#include "common.h"
#include "ddpm_hip.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One thread produces PX horizontally adjacent pixels x 8 channels: per kernel row it loads the PX+2 input
// vectors once and reuses them across the PX outputs, and loads the row's 3x8 weights once.
// STATS (large images, host-checked shapes): every round of 256 work items lies inside one image and covers all channels;
// the workgroup adds the sums (x, x^2) of what it stored per normalisation group and leaves them as one partial pair per
// (image, group, round) in part[N][G][PT][2] for the GroupNorm that follows (ddpm_gn_desc.stat_part).
template <int PX, bool STATS = false>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, bf16* __restrict__ y, int N, int H, int W,
                                                         int C, float* __restrict__ part = nullptr, int G = 0, int PT = 0) {
    __shared__ float s_red[STATS ? 4 * 64 * 8 : 1];
    const int cvs = C / 8, wq = W / PX;
    const long long total = (long long)N * H * wq * cvs;
    // Workgroups are dealt to the 8 XCDs in turn and every XCD has an L2 of its own: neighbouring image rows (whose halo
    // rows are shared) go to workgroups of the SAME XCD, so that the three uses of an input row hit one L2.
#ifndef DW_NO_XCD_MAP
    const unsigned nb = gridDim.x;
    const unsigned lb = nb % 8 == 0 ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
#else
    const unsigned lb = blockIdx.x;
#endif
    // (the host has checked that the item count fits 32 bits: the index arithmetic of an item is a handful of 32-bit operations)
    for (unsigned i = lb * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        const unsigned cv = i % (unsigned)cvs;
        unsigned r = i / (unsigned)cvs;
        const int wx = (int)(r % (unsigned)wq) * PX; r /= (unsigned)wq;
        const int hy = (int)(r % (unsigned)H);
        const long long n = r / (unsigned)H;
        const int c = (int)cv * 8;
        float acc[PX][8];
        if (bias) {
            const f32x4 b0 = *(const f32x4*)(bias + c), b1 = *(const f32x4*)(bias + c + 4);
#pragma unroll
            for (int q = 0; q < PX; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[q][j] = b0[j]; acc[q][4 + j] = b1[j]; }
        } else {
#pragma unroll
            for (int q = 0; q < PX; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = hy + kh - 1;
            if (iy < 0 || iy >= H) continue;
            const bf16* row = x + ((n * H + iy) * W) * C + c;
            // the PX+2 input vectors of this kernel row as raw 16-byte words; only the two outer ones can lie outside the row
            // (their address is clamped and the value replaced by zero)
            u32x4 u[PX + 2];
#pragma unroll
            for (int q = 0; q < PX + 2; ++q) {
                const int ix = wx + q - 1;
                const int ixc = q == 0 ? (ix < 0 ? 0 : ix) : (q == PX + 1 ? (ix >= W ? W - 1 : ix) : ix);
                u[q] = *(const u32x4*)(row + (long long)ixc * C);
                if (q == 0 || q == PX + 1) {
                    const bool ok = ix >= 0 && ix < W;
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[q][k] = ok ? u[q][k] : 0u;
                }
            }
            float wk[3][8];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const f32x4 w0 = *(const f32x4*)(w + (kh * 3 + kw) * C + c), w1 = *(const f32x4*)(w + (kh * 3 + kw) * C + c + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { wk[kw][j] = w0[j]; wk[kw][4 + j] = w1[j]; }
            }
            // every input vector is unpacked once (bf16 -> fp32 is a shift or a mask of the packed word) and feeds the up to
            // three outputs it belongs to; per output the taps are still added in (kh, kw) order
#pragma unroll
            for (int pq = 0; pq < PX + 2; ++pq) {
                float f[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(u[pq][k] << 16); f[2 * k + 1] = __uint_as_float(u[pq][k] & 0xffff0000u); }
#pragma unroll
                for (int kw = 2; kw >= 0; --kw) {
                    const int q = pq - kw;
                    if (q < 0 || q >= PX) continue;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[q][j] += f[j] * wk[kw][j];
                }
            }
        }
        float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_ss[4] = {0.f, 0.f, 0.f, 0.f};       // STATS: sums of this thread's four channel pairs
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[q][j];
            *(bf16x8*)(y + ((n * H + hy) * W + wx + q) * C + c) = o;
            if constexpr (STATS) {
                pair_sums(o, st_s, st_ss);
            }
        }
        if constexpr (STATS) {
            // lanes with the same channel vector (lane % cvs, cvs a power of two <= 64) add up by shuffles, the four waves
            // meet in LDS, one thread per group adds its channel pairs in a fixed order (deterministic)
            for (int o = cvs; o < 64; o <<= 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { st_s[q] += __shfl_xor(st_s[q], o); st_ss[q] += __shfl_xor(st_ss[q], o); }
            }
            const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
            __syncthreads();                               // the previous round's sums have been read
            if (lane < cvs) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { s_red[(wv * cvs + lane) * 8 + q] = st_s[q]; s_red[(wv * cvs + lane) * 8 + 4 + q] = st_ss[q]; }
            }
            __syncthreads();
            const int cpg = C / G;
            if (tid < G) {
                float S = 0.f, SS = 0.f;
                for (int c2 = tid * cpg / 2; c2 < (tid + 1) * cpg / 2; ++c2) {
                    const int col = c2 >> 2, q = c2 & 3;
#pragma unroll
                    for (int wq_ = 0; wq_ < 4; ++wq_) { S += s_red[(wq_ * cvs + col) * 8 + q]; SS += s_red[(wq_ * cvs + col) * 8 + 4 + q]; }
                }
                const long long round = (i - tid) / 256;
                const long long img = round / PT;
                float* o = part + ((img * G + tid) * PT + (round - img * PT)) * 2;
                o[0] = S; o[1] = SS;
            }
        }
    }
}

// partial pairs per (image, group) that ddpm_dwconv3x3_forward_stats writes; 0 where it does not apply (small images, shapes
// whose rounds of 256 work items would straddle images, one-channel groups)
extern "C" int ddpm_dwconv3x3_stat_parts(int N, int H, int W, int C, int groups) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || groups <= 0 || groups > 256 || C % groups) return 0;
    const int cpg = C / groups, cvs = C / 8;
    if (cpg < 2 || (cpg & (cpg - 1)) || cvs > 64 || (cvs & (cvs - 1))) return 0;
    // only where the GroupNorm that follows takes its many-workgroup path (more than 4096 vectors per image and channel slab)
    const int slab = C % 64 == 0 ? 64 : (C % 32 == 0 ? 32 : (C % 16 == 0 ? 16 : 8));
    if ((long long)H * W * (slab / 8) <= 4096) return 0;
    const int px = W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1);
    const long long per_image = (long long)H * (W / px) * cvs;
    if (per_image % 256 || per_image / 256 > (1 << 20)) return 0;
    return (int)(per_image / 256);
}

extern "C" int ddpm_dwconv3x3_forward_stats(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                                            int groups, float* part, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && w && y && part, "dwconv: null pointer");
    const int PT = ddpm_dwconv3x3_stat_parts(N, H, W, C, groups);
    DDPM_REQUIRE(PT > 0, "dwconv: no GroupNorm sums for N=%d %dx%d C=%d groups=%d (ask ddpm_dwconv3x3_stat_parts first)", N, H, W, C, groups);
    DDPM_REQUIRE((long long)N * PT * 256 + 256ll * 4096 < (1ll << 32), "dwconv: %lld work items do not fit the kernel's 32-bit index", (long long)N * PT * 256);
    const int px = W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1);
    long long grid = (long long)N * PT;
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid > 8) grid = (grid + 7) / 8 * 8;
    if (px == 4) hipLaunchKernelGGL((dwconv3x3_kernel<4, true>), dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C, part, groups, PT);
    else if (px == 2) hipLaunchKernelGGL((dwconv3x3_kernel<2, true>), dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C, part, groups, PT);
    else hipLaunchKernelGGL((dwconv3x3_kernel<1, true>), dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C, part, groups, PT);
    return ddpm_check_launch("dwconv3x3 (with GroupNorm sums)");
}

extern "C" int ddpm_dwconv3x3_forward(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                                      ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && w && y, "dwconv: null pointer");
    DDPM_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "dwconv: C (%d) must be a positive multiple of 8", C);
    const int px = W % 4 == 0 ? 4 : (W % 2 == 0 ? 2 : 1);
    const long long total = (long long)N * H * (W / px) * (C / 8);
    DDPM_REQUIRE(total + 256ll * 4096 < (1ll << 32), "dwconv: %lld work items do not fit the kernel's 32-bit index", total);
    long long grid = (total + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    if (grid > 8) grid = (grid + 7) / 8 * 8;              // whole rounds over the XCDs (surplus workgroups find no work)
    if (px == 4) hipLaunchKernelGGL((dwconv3x3_kernel<4, false>), dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C, (float*)nullptr, 0, 0);
    else if (px == 2) hipLaunchKernelGGL((dwconv3x3_kernel<2, false>), dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C, (float*)nullptr, 0, 0);
    else hipLaunchKernelGGL((dwconv3x3_kernel<1, false>), dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x, w, bias, (bf16*)y, N, H, W, C, (float*)nullptr, 0, 0);
    return ddpm_check_launch("dwconv3x3");
}

// ---- depthwise 3x3 on the RAW result of the expansion convolution (inference on large images) ------------------------
// x is the input of a GroupNorm + activation that has not been applied (conv_block.py:67-72 of the reference); affine =
// fp32 [N][C][2] (scale, shift) per image and channel from ddpm_gn_fold_affine.  Applying them in the stencil's loads would
// evaluate the activation 4.5 times per output, so the normalised rows go through LDS instead: a workgroup owns a band of
// DWA_BAND image rows, keeps a ring of activated rows (every input row is normalised once, plus two halo rows per band) and
// produces RS = 1 or 2 output rows per step with the arithmetic and the thread layout of dwconv3x3_kernel<4> (a step is
// exactly one of its rounds of 256 work items, so the GroupNorm sums it leaves have the same layout).  While a step is
// computed from LDS, the next step's raw rows are in flight; one barrier per step.
#define DWA_BAND 16
template <int RS>
__global__ __launch_bounds__(256, 2) void dwconv3x3_affine_kernel(const bf16* __restrict__ x, const float* __restrict__ affine, int act,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               bf16* __restrict__ y, int N, int H, int W, int C,
                                                               float* __restrict__ part, int G, int PT) {
    extern __shared__ __attribute__((aligned(16))) char s_ring[];      // [RING][W][C] bf16, then the reduction scratch
    constexpr int RING = RS == 1 ? 4 : 8;
    __shared__ float s_red[4 * 64 * 8];
    __shared__ __attribute__((aligned(16))) float s_w[9 * 128];      // the taps [9][C] (read from here every step: held in registers they are 72 of them)
    const int tid = threadIdx.x;
    const int cvs = C / 8, wq = W / 4;
    const int row_bytes = W * C * 2, row_vecs = W * cvs;             // RS * row_vecs = 1024: four vectors per thread and step
    const int bands = H / DWA_BAND;
    const int n = blockIdx.x / bands, y_begin = (blockIdx.x - n * bands) * DWA_BAND;
    // this thread's four load slots of a step (vector index tid + 256 u of the RS rows): row in the step, byte offset in the row
    int l_row[4], l_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int v = tid + 256 * u;
        l_row[u] = v / row_vecs;
        l_off[u] = (v - l_row[u] * row_vecs) * 16;
    }
    // (scale, shift) of the eight channels of its load slots (256 % cvs == 0: the same channel vector in every slot)
    float sc[8], sh[8];
    {
        const float* ap = affine + ((long long)n * C + (tid % cvs) * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x4 t4 = *(const f32x4*)(ap + j * 2);
            sc[j] = t4[0]; sh[j] = t4[1]; sc[j + 1] = t4[2]; sh[j + 1] = t4[3];
        }
    }
    const bf16* xn = x + (long long)n * H * W * C;
    u32x4 ld[4];
    bool ld_ok[4];
    // raw vectors of image rows r0 .. r0 + RS - 1 (zeros outside the image: the convolution pads the ACTIVATED tensor)
    auto fetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + l_row[u];
            ld_ok[u] = r >= 0 && r < H;
            const int rc = r < 0 ? 0 : (r >= H ? H - 1 : r);
            ld[u] = *(const u32x4*)((const char*)xn + (long long)rc * row_bytes + l_off[u]);
        }
    };
    auto stash = [&](int r0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + l_row[u];
            float f[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[2 * k] = __uint_as_float(ld[u][k] << 16) * sc[2 * k] + sh[2 * k];
                f[2 * k + 1] = __uint_as_float(ld[u][k] & 0xffff0000u) * sc[2 * k + 1] + sh[2 * k + 1];
            }
            if (act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = act_fwd(f[j], ACT_GELU);
            } else if (act == ACT_SILU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = act_fwd(f[j], ACT_SILU);
            }
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = ld_ok[u] ? (bf16)f[j] : (bf16)0.f;
            asm volatile("" : "+v"(o));            // one vector after the other: four interleaved activation chains do not fit the registers
            *(bf16x8*)(s_ring + ((r + RING) & (RING - 1)) * row_bytes + l_off[u]) = o;
        }
    };
    // the thread's work item of a step: the layout of dwconv3x3_kernel<4> (channel vector fastest, then x group, then row)
    const int cv = tid % cvs, xr = tid / cvs;
    const int wx = (xr % wq) * 4, ry = xr / wq;                      // ry < RS
    const int c = cv * 8;
    float bq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bq[j] = bias ? bias[c + j] : 0.f;
    for (int i = tid; i < 9 * C; i += 256) s_w[i] = w[i];

    // ---- band prologue: rows y_begin - 1 .. y_begin + RS (the first step's inputs); RS + 2 rows = up to two fetch rounds
#pragma unroll 1
    for (int r0 = y_begin - 1; r0 <= y_begin + RS; r0 += RS) { fetch(r0); stash(r0); }
    __syncthreads();
#pragma unroll 1
    for (int y0 = y_begin; y0 < y_begin + DWA_BAND; y0 += RS) {
        const bool more = y0 + RS < y_begin + DWA_BAND;
        if (more) fetch(y0 + RS + 1);                                // rows y0 + RS + 1 .. y0 + 2 RS: new for the next step
        const int hy = y0 + ry;
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[q][j] = bq[j];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const char* row = s_ring + ((hy + kh - 1 + RING) & (RING - 1)) * row_bytes + c * 2;
            float wk[3][8];                                          // the kernel row's taps
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const f32x4 w0 = *(const f32x4*)(s_w + (kh * 3 + kw) * C + c), w1 = *(const f32x4*)(s_w + (kh * 3 + kw) * C + c + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { wk[kw][j] = w0[j]; wk[kw][4 + j] = w1[j]; }
            }
#pragma unroll
            for (int pq = 0; pq < 6; ++pq) {
                const int ix = wx + pq - 1;
                const int ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
                u32x4 uv = *(const u32x4*)(row + ixc * C * 2);
                if (pq == 0 || pq == 5) {
                    const bool ok = ix >= 0 && ix < W;
#pragma unroll
                    for (int k = 0; k < 4; ++k) uv[k] = ok ? uv[k] : 0u;
                }
                float f[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(uv[k] << 16); f[2 * k + 1] = __uint_as_float(uv[k] & 0xffff0000u); }
#pragma unroll
                for (int kw = 2; kw >= 0; --kw) {
                    const int q = pq - kw;
                    if (q < 0 || q >= 4) continue;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[q][j] += f[j] * wk[kw][j];
                }
            }
        }
        float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[q][j];
            *(bf16x8*)(y + (((long long)n * H + hy) * W + wx + q) * C + c) = o;
            pair_sums(o, st_s, st_ss);
        }
        // the step's GroupNorm sums, exactly as dwconv3x3_kernel<4, true> leaves them for its round
        for (int o = cvs; o < 64; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { st_s[q] += __shfl_xor(st_s[q], o); st_ss[q] += __shfl_xor(st_ss[q], o); }
        }
        const int lane = tid & 63, wv = tid >> 6;
        if (lane < cvs) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_red[(wv * cvs + lane) * 8 + q] = st_s[q]; s_red[(wv * cvs + lane) * 8 + 4 + q] = st_ss[q]; }
        }
        if (more) stash(y0 + RS + 1);                                // (slots of rows no step before the next one reads)
        __syncthreads();
        const int cpg = C / G;
        if (tid < G) {
            float S = 0.f, SS = 0.f;
            for (int c2 = tid * cpg / 2; c2 < (tid + 1) * cpg / 2; ++c2) {
                const int col = c2 >> 2, q = c2 & 3;
#pragma unroll
                for (int wq_ = 0; wq_ < 4; ++wq_) { S += s_red[(wq_ * cvs + col) * 8 + q]; SS += s_red[(wq_ * cvs + col) * 8 + 4 + q]; }
            }
            float* o = part + (((long long)n * G + tid) * PT + y0 / RS) * 2;
            o[0] = S; o[1] = SS;
        }
        __syncthreads();                                             // s_red is rewritten by the next step
    }
}

// 1 where ddpm_dwconv3x3_forward_affine takes the shape: rows of 128 or 256 work items (4 pixels x 8 channels each), bands of
// DWA_BAND rows, and the conditions of the GroupNorm sums (ddpm_dwconv3x3_stat_parts)
extern "C" int ddpm_dwconv3x3_affine_supported(int N, int H, int W, int C, int groups) {
    if (ddpm_dwconv3x3_stat_parts(N, H, W, C, groups) <= 0 || W % 4 || H % DWA_BAND) return 0;
    const int items = (W / 4) * (C / 8);
    if ((items != 128 && items != 256) || C > 128) return 0;
    const int rs = 256 / items;
    return ((rs == 1 ? 4 : 8) * W * C * 2 <= 128 * 1024 && (long long)N * (H / DWA_BAND) < (1ll << 31)) ? 1 : 0;
}

extern "C" int ddpm_dwconv3x3_forward_affine(const void* x, const float* affine, int act, const float* w, const float* bias, void* y, int N,
                                             int H, int W, int C, int groups, float* part, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && affine && w && y && part, "dwconv (normalise on load): null pointer");
    DDPM_REQUIRE(act == ACT_NONE || act == ACT_GELU || act == ACT_SILU, "dwconv (normalise on load): activation none / gelu / silu");
    DDPM_REQUIRE(ddpm_dwconv3x3_affine_supported(N, H, W, C, groups), "dwconv (normalise on load): unsupported shape N=%d %dx%d C=%d groups=%d (ask ddpm_dwconv3x3_affine_supported first)", N, H, W, C, groups);
    const int PT = ddpm_dwconv3x3_stat_parts(N, H, W, C, groups);
    const int rs = 256 / ((W / 4) * (C / 8));
    const int lds = (rs == 1 ? 4 : 8) * W * C * 2;
    const dim3 grid((unsigned)(N * (H / DWA_BAND)));
    if (rs == 1) {
        static bool attr = false;
        if (!attr) { if (hipFuncSetAttribute((const void*)dwconv3x3_affine_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return ddpm_check_launch("dwconv: raise dynamic LDS limit"); attr = true; }
        hipLaunchKernelGGL((dwconv3x3_affine_kernel<1>), grid, dim3(256), lds, stream, (const bf16*)x, affine, act, w, bias, (bf16*)y, N, H, W, C, part, groups, PT);
    } else {
        static bool attr = false;
        if (!attr) { if (hipFuncSetAttribute((const void*)dwconv3x3_affine_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess) return ddpm_check_launch("dwconv: raise dynamic LDS limit"); attr = true; }
        hipLaunchKernelGGL((dwconv3x3_affine_kernel<2>), grid, dim3(256), lds, stream, (const bf16*)x, affine, act, w, bias, (bf16*)y, N, H, W, C, part, groups, PT);
    }
    return ddpm_check_launch("dwconv3x3 (normalise on load)");
}

// part[b][t][c]: t < 9: sum over the block's pixel strip of dy[n,h,w,c] * x[n,h+kh-1,w+kw-1,c]; t = 9: sum of dy.
// Thread layout: 256 threads = (256/cvs) pixel lanes x cvs channel vectors (cvs = C/8, a power of two <= 64).
// Lanes of a wave that share a channel vector are combined with shuffles, the four waves through LDS; the
// workgroup then stores its [10][C] partial with plain stores (a second pass sums over workgroups).
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                               float* __restrict__ part, int N, int H, int W, int C,
                                                               int pix_per_block) {
    extern __shared__ float s_part[];      // [4 waves][80][lo]
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
    // (wx, hy) of the thread's first pixel by division, then advanced by pstep pixels per iteration with carries: two 64-bit
    // divisions per pixel cost about as much as the pixel's 80 multiply-adds
    long long pix = p0 + prow;
    int wx = (int)(pix % W), hy = (int)((pix / W) % H);
    const int step_y = pstep / W, step_x = pstep - step_y * W;
    for (; pix < p1; pix += pstep, wx += step_x, hy += step_y) {
        if (wx >= W) { wx -= W; ++hy; }
        while (hy >= H) hy -= H;
        const bf16x8 g = *(const bf16x8*)(dy + pix * C + c);
        bf16x8 v[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iy = hy + kh - 1, ix = wx + kw - 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[kh * 3 + kw][j] = (bf16)0.f;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[kh * 3 + kw] = *(const bf16x8*)(x + (pix + (kh - 1) * W + (kw - 1)) * C + c);
            }
        float gf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { gf[j] = (float)g[j]; sb[j] += gf[j]; }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][j] += gf[j] * (float)v[t][j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lo = cvs < 64 ? cvs : 64;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = t < 9 ? acc[t < 9 ? t : 0][j] : sb[j];
            for (int o = 32; o >= lo; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < lo) s_part[(wave * 80 + t * 8 + j) * lo + lane] = v;
        }
    __syncthreads();
    // waves hold disjoint channel vectors when cvs > 64 is impossible (cvs <= 64): every wave covers vectors
    // lane % cvs, so the same `lo` vectors appear in all four waves -> sum the four copies
    float* out = part + (long long)blockIdx.x * 10 * C;
    for (int i = threadIdx.x; i < 80 * lo; i += 256) {
        const int k = i / lo, l = i - k * lo;           // k = t*8 + j, l = channel vector
        const float v = s_part[(0 * 80 + k) * lo + l] + s_part[(1 * 80 + k) * lo + l] + s_part[(2 * 80 + k) * lo + l] + s_part[(3 * 80 + k) * lo + l];
        out[(k >> 3) * C + l * 8 + (k & 7)] = v;
    }
}

static void dw_wgrad_split(int N, int H, int W, int C, int* blocks, int* ppb) {
    const long long npix = (long long)N * H * W;
    const int cvs = C / 8;
    long long b = (npix * cvs + 256 * 8 - 1) / (256 * 8);     // ~8 pixels per thread: partials stay ~10 % of the input bytes
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    *ppb = (int)((npix + b - 1) / b);
    *blocks = (int)((npix + *ppb - 1) / *ppb);
}

extern "C" int ddpm_dwconv3x3_wgrad_blocks(int N, int H, int W, int C) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return 0;
    int blocks, ppb;
    dw_wgrad_split(N, H, W, C, &blocks, &ppb);
    return blocks;
}

extern "C" int ddpm_dwconv3x3_wgrad(const void* x, const void* dy, float* part, int N, int H, int W, int C, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(x && dy && part, "dwconv_wgrad: null pointer");
    const int cvs = C / 8;
    DDPM_REQUIRE(N > 0 && H > 0 && W > 0 && C % 8 == 0 && cvs >= 1 && cvs <= 64 && (cvs & (cvs - 1)) == 0,
                 "dwconv_wgrad: C (%d) must be 8 * a power of two <= 512", C);
    int blocks, ppb;
    dw_wgrad_split(N, H, W, C, &blocks, &ppb);
    const size_t lds = (size_t)4 * 80 * cvs * sizeof(float);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)dwconv3x3_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ddpm_check_launch("dwconv_wgrad: raise dynamic LDS limit");
    }
    hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, (const bf16*)x, (const bf16*)dy, part, N, H,
                       W, C, ppb);
    return ddpm_check_launch("dwconv3x3_wgrad");
}
