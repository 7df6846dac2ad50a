// Weight gradient of the thin 3x3 stride-1 convolutions at the image ends of large batches: the stem (16 stored input
// channels -> 32), the first-stage and feature convolutions (32 -> 32) and the head (32 -> at most 16 stored output channels)
// (unets.py:78-85,160-167 of the reference, backward).
//
//   dW[ky][kx][co][ci] = sum over positions (n, y, x) of  dY[n][y][x][co] * X[n][y + ky - 1][x + kx - 1][ci]
//
// Hundreds of thousands of positions against a 9 x 32 x 32 result: the tap-list kernel of conv_wgrad.hip spends its time on
// per-tile tables and barriers here, and these problems are the last leaf work of a training step, which nothing hides.
// One workgroup per CU walks bands of 256 positions (whole image rows).  A band's dY rows and its zero-haloed input patch sit
// in LDS; every wave multiplies its 64 positions of the band into nine 32 x 32 fp32 accumulator tiles that live in registers
// for the whole launch (both operands are pixel-major, i.e. [k][row] for the matrix instruction: transposing LDS reads), and the
// next band's global loads are in flight meanwhile.  At the end the four waves' tiles meet in LDS, tap by tap, and the
// workgroup stores one partial weight block (`part`, folded by ddpm_reduce_rows) or adds it with atomics.
// Channels beyond the stored ones (input 16 .. 31 of the stem, output 16 .. 31 of the head) are computed on whatever follows
// in LDS and never stored: a column of the product depends on its own column of the operand only.
#include "common.h"
#include "ddpm_hip.h"

int ddpm_conv_pipe_mode_now();            // conv_pipe.hip: 0 by problem size, 1 wherever the shape allows, -1 never

namespace {

constexpr int BAND = 256;                 // positions per band
constexpr int MAXP = 8, MAXD = 4;         // 16-byte vectors per thread and band: patch, dY

struct WcArgs {
    const bf16* x; const bf16* dy; const float* in_add;
    float* dw; float* part; long long dw_numel;
    int N, H, W, Cout, bands, G;          // bands = N * H / (BAND / W); G = workgroups
};

__device__ __forceinline__ bf16x8 tr_pair(const char* p0, const char* p1) {
    // two transposed 4 x 16 block reads -> the 8 k-values of this lane's row / column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

// CI: stored input channels (16 or 32); DC: stored channels of dY (16 or 32); W: image width (32, 64 or 128)
template <int CI, int DC, int W>
__global__ __launch_bounds__(256) void conv_wgrad_c32_kernel(const WcArgs a) {
    constexpr int R = BAND / W;                         // image rows per band
    constexpr int PW = W + 2;                           // patch row length in pixels (one zero column on either side)
    constexpr int PS = CI * 2 + 16, DS = DC * 2 + 16;   // pixel strides in LDS (bytes): consecutive pixels land 12 / 20 banks apart
    constexpr int VPP = CI / 8, DV = DC / 8;            // 16-byte vectors per pixel
    constexpr int NPV = (R + 2) * W * VPP, NDV = BAND * DV;
    constexpr int PATCH_BYTES = (R + 2) * PW * PS + 64;  // + slack: the stem's reads of channels 16 .. 31 run past the last pixel
    static_assert(NPV <= MAXP * 256 && NDV <= MAXD * 256, "slots");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* s_patch = smem;
    char* s_dy = smem + PATCH_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // the halo columns (and the slack) stay zero for the whole launch; the rows between them are rewritten per band
    for (int i = tid; i < (PATCH_BYTES + BAND * DS + 64) / 16; i += 256) *(f32x4*)(smem + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int kq = 8 * (grp >> 1) + q;                  // k row (position) this lane supplies to a transposing read; + 4 for the second
    const int colb = (16 * (grp & 1) + 4 * pp) * 2;     // byte offset of its 4 channels inside a pixel

    const int bands_per_img = a.H / R;
    bf16x8 pv[MAXP], dv[MAXD];
    auto fetch = [&](int band) {
        const int n = band / bands_per_img, r0 = (band - n * bands_per_img) * R;
        const bf16* xi = a.x + (long long)n * a.H * W * CI;
        const bf16* di = a.dy + ((long long)n * a.H + r0) * W * DC;
        const float* ad = a.in_add ? a.in_add + (long long)n * CI : nullptr;
#pragma unroll
        for (int u = 0; u < MAXP; ++u) {
            const int it = tid + u * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[u][j] = (bf16)0.f;
            if (it >= NPV) continue;
            const int pr = it / (W * VPP), rem = it - pr * (W * VPP);
            const int y = r0 - 1 + pr;
            if (y < 0 || y >= a.H) continue;
            pv[u] = ld_at<bf16x8>(xi, (unsigned)((y * W) * CI + rem * 8) * 2);
            if (ad) {
                const int c = (rem % VPP) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[u][j] = (bf16)((float)pv[u][j] + ad[c + j]);
            }
        }
#pragma unroll
        for (int u = 0; u < MAXD; ++u) {
            const int it = tid + u * 256;
            if (it < NDV) dv[u] = ld_at<bf16x8>(di, (unsigned)it * 16);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < MAXP; ++u) {
            const int it = tid + u * 256;
            if (it >= NPV) continue;
            const int pr = it / (W * VPP), rem = it - pr * (W * VPP);
            const int px = rem / VPP, part = rem - px * VPP;
            *(bf16x8*)(s_patch + (pr * PW + px + 1) * PS + part * 16) = pv[u];
        }
#pragma unroll
        for (int u = 0; u < MAXD; ++u) {
            const int it = tid + u * 256;
            if (it >= NDV) continue;
            const int p = it / DV, part = it - p * DV;
            *(bf16x8*)(s_dy + p * DS + part * 16) = dv[u];
        }
    };

    int band = blockIdx.x;
    if (band < a.bands) fetch(band);
    for (; band < a.bands; band += a.G) {
        __syncthreads();                                // the previous band has been read (first trip: the zero fill is complete)
        stash();
        __syncthreads();
        if (band + a.G < a.bands) fetch(band + a.G);    // in flight while this band is multiplied
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int p0 = wave * 64 + ks * 16;         // 16 consecutive positions of one image row
            const int r = p0 / W, c = p0 - r * W;
            const char* dp = s_dy + (p0 + kq) * DS + colb;
            const bf16x8 av = tr_pair(dp, dp + 4 * DS);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const char* xp = s_patch + ((r + kh) * PW + c + kw + kq) * PS + colb;
                    const bf16x8 bv = tr_pair(xp, xp + 4 * PS);
                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[kh * 3 + kw], 0, 0, 0);
                }
        }
    }

    // the four waves' tiles, tap by tap: element r of a lane = output channel (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), input
    // channel lane & 31
    float* red = (float*)smem;                          // [wave][32][32]
    float* out = a.part ? a.part + (long long)blockIdx.x * a.dw_numel : a.dw;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            red[wave * 1024 + co * 32 + (lane & 31)] = acc[t][r];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * 256, co = e >> 5, ci = e & 31;
            const float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
            if (co < a.Cout && ci < CI) {
                float* o = out + ((long long)t * a.Cout + co) * CI + ci;
                if (a.part) *o = v; else atomicAdd(o, v);
            }
        }
    }
}

template <int CI, int DC, int W>
int launch(const WcArgs& a, hipStream_t stream) {
    constexpr int R = BAND / W;
    constexpr int lds = (R + 2) * (W + 2) * (CI * 2 + 16) + 64 + BAND * (DC * 2 + 16) + 64;
    static_assert(lds >= 4 * 1024 * 4 && lds <= 64 * 1024, "LDS");
    hipLaunchKernelGGL((conv_wgrad_c32_kernel<CI, DC, W>), dim3((unsigned)a.G), dim3(256), lds, stream, a);
    return ddpm_check_launch("conv_wgrad_c32");
}

// shape part of the decision (the descriptor's pointers are only tested for presence)
bool plan(const ddpm_wgrad_desc* d, WcArgs& a) {
    const int mode = ddpm_conv_pipe_mode_now();
    if (mode < 0) return false;
    if (d->kind != DDPM_CONV_3X3_S1 || d->tiles_per_split > 0 || d->C1 || d->in1 || d->in_mask) return false;
    if ((d->C0 != 16 && d->C0 != 32) || (d->dy_cs != 16 && d->dy_cs != 32) || d->Cout > d->dy_cs || d->Cout <= 0) return false;
    if (d->C0 == 16 && d->dy_cs == 16) return false;
    if (d->IW != 32 && d->IW != 64 && d->IW != 128) return false;
    const int R = BAND / d->IW;
    if (d->N <= 0 || d->IH <= 0 || d->IH % R) return false;
    const long long bands = (long long)d->N * (d->IH / R);
    if (bands * BAND * 32 >= (1ll << 31)) return false;               // 32-bit element offsets inside an image are enough; be safe
    // by size: from one band per CU on (65 536 positions); below, the grouped launch takes the problem with all the others
    if (mode == 0 && bands < 256) return false;
    a.N = d->N; a.H = d->IH; a.W = d->IW; a.Cout = d->Cout;
    a.bands = (int)bands; a.G = bands < 256 ? (int)bands : 256;
    a.dw_numel = 9ll * d->Cout * d->C0;
    return true;
}

}  // namespace

// Number of `part` rows this kernel writes for the problem, or 0 when it does not take it.
int ddpm_conv_wgrad_c32_splits(const ddpm_wgrad_desc* d) {
    WcArgs a;
    return plan(d, a) ? a.G : 0;
}

// Returns 1 when the problem was launched here, 0 when the caller should use another kernel, < 0 on error.
int ddpm_conv_wgrad_c32_try(const ddpm_wgrad_desc* d, hipStream_t stream) {
    WcArgs a;
    if (!plan(d, a)) return 0;
    a.x = (const bf16*)d->in0; a.dy = (const bf16*)d->dy; a.in_add = d->in_add; a.dw = d->dw; a.part = d->part;
    int rc = DDPM_EINVAL;
#define WC_CASE(CI, DC, W) if (d->C0 == CI && d->dy_cs == DC && d->IW == W) rc = launch<CI, DC, W>(a, stream);
    WC_CASE(16, 32, 32) WC_CASE(16, 32, 64) WC_CASE(16, 32, 128)
    WC_CASE(32, 32, 32) WC_CASE(32, 32, 64) WC_CASE(32, 32, 128)
    WC_CASE(32, 16, 32) WC_CASE(32, 16, 64) WC_CASE(32, 16, 128)
#undef WC_CASE
    return rc == DDPM_OK ? 1 : -1;
}
