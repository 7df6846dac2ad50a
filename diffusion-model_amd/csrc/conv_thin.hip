// 3x3 stride-1 convolution with very few OUTPUT channels (the UNet's image head: 32 -> 1 or 3 channels, tanh, fp32 NCHW).
//
// On the implicit-GEMM kernel such a layer computes a 32-wide MFMA tile to keep one column of it and is bound by the
// staging of its input; here a thread owns one output pixel, the input patch of a 8 x 32 pixel tile (with halo) sits in
// LDS, and the 288 (x Cout) multiply-adds per pixel run on the vector ALUs as packed-bf16 dot products against weights held
// in scalar registers.
// Same operands as ddpm_conv_forward (packed bf16 weights [9][w_rows][Cin], fp32 bias), products of bf16 values
// accumulated in fp32 like the MFMA path (the summation order differs).
#include "conv_geom.h"
#include "ddpm_hip.h"

#define THIN_TH 8
#define THIN_TW 32
#define THIN_MAXCO 4

struct ThinArgs {
    const bf16* in; const bf16* w; const float* bias; float* out;
    int N, H, W, Cin, Cout, w_rows, act;
};

__global__ __launch_bounds__(256) void conv_thin_out_kernel(const ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int PS = a.Cin * 2 + 16;                                // patch pixel stride in bytes (16 B pad against bank conflicts)
    constexpr int PW = THIN_TW + 2, PH = THIN_TH + 2;
    char* s_patch = smem;                                          // [PH][PW][PS]
    const int tid = threadIdx.x;
    const int tiles_w = (a.W + THIN_TW - 1) / THIN_TW, tiles_h = (a.H + THIN_TH - 1) / THIN_TH;
    int t = blockIdx.x;
    const int twi = t % tiles_w; t /= tiles_w;
    const int thi = t % tiles_h;
    const int n = t / tiles_h;
    const int y0 = thi * THIN_TH, x0 = twi * THIN_TW;
    const int vpp = a.Cin / 8;                                     // 16-byte vectors per pixel

    const int items = PH * PW * vpp;
    for (int base = tid; base < items; base += 256 * 6) {
        bf16x8 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = base + u * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
            if (it < items) {
                const int p = it / vpp, part = it - p * vpp;
                const int py = p / PW, px = p - py * PW;
                const int iy = y0 + py - 1, ix = x0 + px - 1;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    v[u] = *(const bf16x8*)(a.in + (((long long)n * a.H + iy) * a.W + ix) * a.Cin + part * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = base + u * 256;
            if (it < items) {
                const int p = it / vpp, part = it - p * vpp;
                *(bf16x8*)(s_patch + p * PS + part * 16) = v[u];
            }
        }
    }
    __syncthreads();
    // (input widths the matrix-core variant below does not take)  v_dot2c_f32_bf16 multiplies two channel pairs of packed bf16
    // and accumulates in fp32: no unpacking of the patch, two multiply-adds per instruction, products exact as on the matrix cores.
    const int ty = tid / THIN_TW, tx = tid % THIN_TW;
    float acc[THIN_MAXCO];
#pragma unroll
    for (int co = 0; co < THIN_MAXCO; ++co) acc[co] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const char* pp = s_patch + ((ty + dy) * PW + tx + dx) * PS;
        const bf16* wt = a.w + (long long)tap * a.w_rows * a.Cin;
        for (int part = 0; part < vpp; ++part) {
            const bf16x8 xv = *(const bf16x8*)(pp + part * 16);
#pragma unroll
            for (int co = 0; co < THIN_MAXCO; ++co) {
                if (co < a.Cout) {
                    const bf16x8 wv = *(const bf16x8*)(wt + co * a.Cin + part * 8);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bf16x2 wp = {wv[2 * q], wv[2 * q + 1]}, xp = {xv[2 * q], xv[2 * q + 1]};
                        acc[co] = __builtin_amdgcn_fdot2_f32_bf16(wp, xp, acc[co], false);
                    }
                }
            }
        }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < a.H && ox < a.W) {
#pragma unroll
        for (int co = 0; co < THIN_MAXCO; ++co)
            if (co < a.Cout)
                a.out[(((long long)n * a.Cout + co) * a.H + oy) * a.W + ox] = act_fwd(acc[co] + (a.bias ? a.bias[co] : 0.f), a.act);
    }
}

// The same on the matrix cores, for 16 / 32 / 64 input channels.  Measured on the vector-ALU kernel above (128x128, 256 images):
// 94 us go to staging the patches, 150 us to the arithmetic, and that arithmetic is bound by the weight operands (every lane
// needs the same 9 x Cin x Cout values: as broadcast loads they keep the vector memory path busy, as LDS reads the LDS).  A
// 32 x 32 x 16 MFMA with the weights as its A operand (rows = output channels, rows >= Cout zero) wastes 29 of 32 rows and is
// still several times cheaper: the weights live in registers for the life of the wave, the patch is read once per tap and
// 16 channels in operand layout, and a lane ends up with the Cout results of ITS pixel in the first accumulator registers,
// so the fp32 NCHW stores of a wave are whole 128-byte rows.
template <int KS>
__global__ __launch_bounds__(256, KS <= 2 ? 4 : 2) void conv_thin_out_mfma_kernel(const ThinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CIN = KS * 16, PS = CIN * 2 + 16, vpp = CIN / 8;
    constexpr int PW = THIN_TW + 2, PH = THIN_TH + 2;
    char* s_patch = smem;                                          // [PH][PW][PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (a.W + THIN_TW - 1) / THIN_TW, tiles_h = (a.H + THIN_TH - 1) / THIN_TH;
    int t = blockIdx.x;
    const int twi = t % tiles_w; t /= tiles_w;
    const int thi = t % tiles_h;
    const int n = t / tiles_h;
    const int y0 = thi * THIN_TH, x0 = twi * THIN_TW;

    constexpr int items = PH * PW * vpp;
    for (int base = tid; base < items; base += 256 * 6) {
        bf16x8 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = base + u * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
            if (it < items) {
                const int p = it / vpp, part = it - p * vpp;
                const int py = p / PW, px = p - py * PW;
                const int iy = y0 + py - 1, ix = x0 + px - 1;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    v[u] = *(const bf16x8*)(a.in + (((long long)n * a.H + iy) * a.W + ix) * CIN + part * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = base + u * 256;
            if (it < items) {
                const int p = it / vpp, part = it - p * vpp;
                *(bf16x8*)(s_patch + p * PS + part * 16) = v[u];
            }
        }
    }
    // weights: lane = output channel (lane & 31), 8 input channels at 16 * ks + 8 * (lane >> 5) of every tap
    bf16x8 wf[9][KS];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[tap][ks][j] = (bf16)0.f;
            if ((lane & 31) < a.Cout)
                wf[tap][ks] = *(const bf16x8*)(a.w + ((long long)tap * a.w_rows + (lane & 31)) * CIN + ks * 16 + (lane >> 5) * 8);
        }
    float bias[THIN_MAXCO];
#pragma unroll
    for (int co = 0; co < THIN_MAXCO; ++co) bias[co] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
    __syncthreads();
    // a wave owns two of the tile's eight rows; B operand = the 32 pixels of a row (column = pixel), shifted by the tap
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int ty = wave * 2 + rr;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const char* pp = s_patch + ((ty + dy) * PW + (lane & 31) + dx) * PS + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 xv = *(const bf16x8*)(pp + ks * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap][ks], xv, acc, 0, 0, 0);
            }
        }
        // accumulator element r of a lane < 32 = output channel r (0 .. 3) of pixel (ty, lane)
        const int oy = y0 + ty, ox = x0 + lane;
        if (lane < 32 && oy < a.H && ox < a.W) {
#pragma unroll
            for (int co = 0; co < THIN_MAXCO; ++co)
                if (co < a.Cout) a.out[(((long long)n * a.Cout + co) * a.H + oy) * a.W + ox] = act_fwd(acc[co] + bias[co], a.act);
        }
    }
}

// The mirror case: very few INPUT channels that matter (the stem: the image's 1 or 3 channels, stored zero-padded to 16;
// and the head's data-gradient, whose input is the gradient of those channels), 32 output channels.  The patch keeps only the
// channels the caller declares as used (C0_used, at most 4); the products run on the matrix cores with K = taps x 4 channels.
struct ThinInArgs {
    const bf16* in; const bf16* w; const float* bias; bf16* out; bf16* out2;
    int N, H, W, Cin, Cused, Cout, out_cs, w_rows, act, act2;
};

__global__ __launch_bounds__(256) void conv_thin_in_kernel(const ThinInArgs a) {
    constexpr int PW = THIN_TW + 2, PH = THIN_TH + 2;
    __shared__ __attribute__((aligned(16))) bf16 s_patch[PH * PW][4];
    const int tid = threadIdx.x;
    const int tiles_w = (a.W + THIN_TW - 1) / THIN_TW, tiles_h = (a.H + THIN_TH - 1) / THIN_TH;
    int t = blockIdx.x;
    const int twi = t % tiles_w; t /= tiles_w;
    const int thi = t % tiles_h;
    const int n = t / tiles_h;
    const int y0 = thi * THIN_TH, x0 = twi * THIN_TW;
    // K = (tap, channel) pairs: k = 4 * tap + ci, 36 of them padded to three 16-wide steps.  A operand (row = output channel):
    // the 8 values of a lane are the first four input channels of two neighbouring taps (8 contiguous bytes each in the
    // weight tensor), requested here, ahead of the patch, and cleared where the channel is not in use
    const int lane = tid & 63, wave = tid >> 6;
    bf16x8 wf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tap = ks * 4 + (lane >> 5) * 2 + h, co = lane & 31;
            bf16x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (bf16)0.f;
            if (tap < 9 && co < a.Cout) v = *(const bf16x4*)(a.w + ((long long)tap * a.w_rows + co) * a.Cin);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[ks][h * 4 + j] = j < a.Cused ? v[j] : (bf16)0.f;
        }
    {
        // 340 patch pixels, 256 threads: both of a thread's loads are in flight before the first is stored
        bf16x4 v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = tid + u * 256;
            const int py = p / PW, px = p - py * PW;
            const int iy = y0 + py - 1, ix = x0 + px - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[u][j] = (bf16)0.f;
            if (p < PH * PW && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                v[u] = *(const bf16x4*)(a.in + (((long long)n * a.H + iy) * a.W + ix) * a.Cin);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + u * 256 < PH * PW) *(bf16x4*)s_patch[tid + u * 256] = v[u];
    }
    __syncthreads();
    // B operand (column = pixel): the 4-channel vectors of the same two taps.  Three matrix instructions per 32 pixels replace
    // 27 .. 36 x 32 multiply-adds per pixel on the vector ALUs.
    // results through LDS, so that consecutive lanes store consecutive 16-byte pieces
    __shared__ __attribute__((aligned(16))) char s_out[256 * 80];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int ty = wave * 2 + rr, tx = lane & 31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            bf16x8 xv;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tap = ks * 4 + (lane >> 5) * 2 + h;
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (bf16)0.f;
                if (tap < 9) v = *(const bf16x4*)s_patch[(ty + tap / 3) * PW + tx + tap % 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[h * 4 + j] = v[j];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xv, acc, 0, 0, 0);
        }
        // accumulator element r = output channel (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) of pixel (ty, tx)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = q * 8 + 4 * (lane >> 5);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[4 * q + j] + ((a.bias && c0 + j < a.Cout) ? a.bias[c0 + j] : 0.f);
                o[j] = (bf16)(a.act == ACT_NONE ? v : act_fwd(v, a.act));      // the stem hands its activated result out as out2
            }
            *(bf16x4*)(s_out + (ty * THIN_TW + tx) * 80 + c0 * 2) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int it = tid + k * 256, px = it >> 2, q = it & 3;
        const int oy = y0 + px / THIN_TW, ox = x0 + px % THIN_TW;
        if (oy >= a.H || ox >= a.W) continue;
        const bf16x8 o = *(const bf16x8*)(s_out + px * 80 + q * 16);
        const long long e = (((long long)n * a.H + oy) * a.W + ox) * a.out_cs + q * 8;
        if (q * 8 + 8 <= a.Cout) *(bf16x8*)(a.out + e) = o;
        else
            for (int j = 0; j < a.Cout - q * 8; ++j) a.out[e + j] = o[j];
        if (a.out2 && q * 8 + 8 <= a.Cout) {              // second result: act2 of the (bf16) first one, Cout % 8 == 0
            bf16x8 y;
            if (a.act2 == ACT_GELU) {             // decided outside the element loop: a switch inside is a branch per element
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)o[j], ACT_GELU);
            } else if (a.act2 == ACT_SILU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)o[j], ACT_SILU);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)o[j], a.act2);
            }
            *(bf16x8*)(a.out2 + e) = y;
        }
    }
}

static int conv_thin_in_try(const ddpm_conv_desc* d, hipStream_t stream) {
    if (d->kind != CONV_3X3_S1 || d->out_mode != OUT_BF16_NHWC || d->C1 != 0 || d->C0_used < 1 || d->C0_used > 4) return 0;
    if (d->in_add || d->in_mask || d->out_mask || d->res || d->sum_out || d->gn_gamma) return 0;
    if (d->out2 && (d->Cout % 8 || d->act != 0)) return 0;
    if (d->C0 % 4 || d->Cout > 32 || d->CoutW != 32 || d->out_cs % 8) return 0;
    ThinInArgs a;
    a.in = (const bf16*)d->in0; a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (bf16*)d->out;
    a.N = d->N; a.H = d->IH; a.W = d->IW; a.Cin = d->C0; a.Cused = d->C0_used; a.Cout = d->Cout; a.out_cs = d->out_cs;
    a.w_rows = d->w_rows; a.act = d->act; a.out2 = (bf16*)d->out2; a.act2 = d->act2;
    const long long tiles = (long long)d->N * ((d->IH + THIN_TH - 1) / THIN_TH) * ((d->IW + THIN_TW - 1) / THIN_TW);
    if (tiles > 0x7fffffffll) return 0;
    hipLaunchKernelGGL(conv_thin_in_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, a);
    return ddpm_check_launch("conv_thin_in") == DDPM_OK ? 1 : -1;
}

// 3x3 stride 1 from 32 to 32 channels (the last feature convolution in front of the head, its data-gradient, and the
// full-resolution convolutions of the first stage): the same skeleton with a full operand.  The general tap-list kernel spends
// most of such a launch on its per-tile bookkeeping (pixel table, weights staged to LDS for every 128-position tile); here the
// nine 32 x 32 weight slices sit in registers as A operands (72 registers), a workgroup stages one 8 x 32 pixel patch, and each
// wave runs 18 matrix instructions per row of 32 pixels.  Epilogue as for the stem: bias, activation, optional residual,
// optional second (activated) result, 16-byte NHWC stores through LDS.
struct C32Args {
    const bf16* in; const bf16* w; const float* bias; bf16* out; bf16* out2; const bf16* res;
    int N, H, W, out_cs, w_rows, act, act2;
};

__global__ __launch_bounds__(256, 4) void conv3x3_c32_kernel(const C32Args a) {
    constexpr int CIN = 32, PS = CIN * 2 + 16, vpp = CIN / 8;
    constexpr int PW = THIN_TW + 2, PH = THIN_TH + 2;
    __shared__ __attribute__((aligned(16))) char smem[PH * PW * PS];      // the patch; afterwards the output tile [256][80]
    char* s_patch = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_w = (a.W + THIN_TW - 1) / THIN_TW, tiles_h = (a.H + THIN_TH - 1) / THIN_TH;
    int t = blockIdx.x;
    const int twi = t % tiles_w; t /= tiles_w;
    const int thi = t % tiles_h;
    const int n = t / tiles_h;
    const int y0 = thi * THIN_TH, x0 = twi * THIN_TW;

    constexpr int items = PH * PW * vpp;
    for (int base = tid; base < items; base += 256 * 6) {
        bf16x8 v[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = base + u * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
            if (it < items) {
                const int p = it / vpp, part = it - p * vpp;
                const int py = p / PW, px = p - py * PW;
                const int iy = y0 + py - 1, ix = x0 + px - 1;
                if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                    v[u] = *(const bf16x8*)(a.in + (((long long)n * a.H + iy) * a.W + ix) * CIN + part * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int it = base + u * 256;
            if (it < items) {
                const int p = it / vpp, part = it - p * vpp;
                *(bf16x8*)(s_patch + p * PS + part * 16) = v[u];
            }
        }
    }
    bf16x8 wf[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            wf[tap][ks] = *(const bf16x8*)(a.w + ((long long)tap * a.w_rows + (lane & 31)) * CIN + ks * 16 + (lane >> 5) * 8);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int ty = wave * 2 + rr;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rr][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const char* pp = s_patch + ((ty + dy) * PW + (lane & 31) + dx) * PS + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 xv = *(const bf16x8*)(pp + ks * 32);
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap][ks], xv, acc[rr], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                   // every wave is done with the patch: its LDS becomes the output tile
    char* s_out = smem;
    // accumulator element r = output channel (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) of pixel (ty, lane & 31)
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int ty = wave * 2 + rr, tx = lane & 31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = q * 8 + 4 * (lane >> 5);
            bf16x4 o;
            if (a.act == ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[rr][4 * q + j] + (a.bias ? a.bias[c0 + j] : 0.f));
            } else if (a.act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)act_fwd(acc[rr][4 * q + j] + (a.bias ? a.bias[c0 + j] : 0.f), ACT_GELU);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)act_fwd(acc[rr][4 * q + j] + (a.bias ? a.bias[c0 + j] : 0.f), a.act);
            }
            *(bf16x4*)(s_out + (ty * THIN_TW + tx) * 80 + c0 * 2) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int it = tid + k * 256, px = it >> 2, q = it & 3;
        const int oy = y0 + px / THIN_TW, ox = x0 + px % THIN_TW;
        if (oy >= a.H || ox >= a.W) continue;
        bf16x8 o = *(const bf16x8*)(s_out + px * 80 + q * 16);
        const long long e = (((long long)n * a.H + oy) * a.W + ox) * a.out_cs + q * 8;
        if (a.res) {
            const bf16x8 rv = *(const bf16x8*)(a.res + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)o[j] + (float)rv[j]);
        }
        *(bf16x8*)(a.out + e) = o;
        if (a.out2) {                                  // second result: act2 of the (bf16) first one
            bf16x8 y;
            if (a.act2 == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)o[j], ACT_GELU);
            } else if (a.act2 == ACT_SILU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)o[j], ACT_SILU);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)o[j], a.act2);
            }
            *(bf16x8*)(a.out2 + e) = y;
        }
    }
}

static int conv_c32_try(const ddpm_conv_desc* d, hipStream_t stream) {
    if (d->kind != CONV_3X3_S1 || d->out_mode != OUT_BF16_NHWC || d->C0 != 32 || d->C1 != 0 || d->Cout != 32 || d->CoutW != 32) return 0;
    if (d->in_add || d->in_mask || d->out_mask || d->sum_out || d->gn_gamma || d->gnb_raw || d->gn_part || d->in_affine || d->C0_used) return 0;
    if (!d->out || d->out_cs % 8 || d->out_cs < 32 || (d->out2 && (d->act != 0 || d->res))) return 0;
    if (ACT_HAS_BETA(d->act) || ACT_HAS_BETA(d->act2)) return 0;
    C32Args a;
    a.in = (const bf16*)d->in0; a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (bf16*)d->out; a.out2 = (bf16*)d->out2;
    a.res = (const bf16*)d->res;
    a.N = d->N; a.H = d->IH; a.W = d->IW; a.out_cs = d->out_cs; a.w_rows = d->w_rows; a.act = d->act; a.act2 = d->act2;
    const long long tiles = (long long)d->N * ((d->IH + THIN_TH - 1) / THIN_TH) * ((d->IW + THIN_TW - 1) / THIN_TW);
    if (tiles > 0x7fffffffll) return 0;
    hipLaunchKernelGGL(conv3x3_c32_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, a);
    return ddpm_check_launch("conv3x3_c32") == DDPM_OK ? 1 : -1;
}

// Returns 1 when the problem was launched here, 0 when the caller should use the general kernel, < 0 on error.
int ddpm_conv_thin_try(const ddpm_conv_desc* d, hipStream_t stream) {
    if (const int rc = conv_thin_in_try(d, stream)) return rc;
    if (const int rc = conv_c32_try(d, stream)) return rc;
    if (d->kind != CONV_3X3_S1 || d->out_mode != OUT_F32_NCHW || d->Cout > THIN_MAXCO || d->C1 != 0) return 0;
    if (d->in_add || d->in_mask || d->out_mask || d->res || d->sum_out || d->gn_gamma) return 0;
    if (d->C0 % 8 || d->C0 > 64) return 0;
    ThinArgs a;
    a.in = (const bf16*)d->in0; a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (float*)d->out;
    a.N = d->N; a.H = d->IH; a.W = d->IW; a.Cin = d->C0; a.Cout = d->Cout; a.w_rows = d->w_rows; a.act = d->act;
    const size_t lds = (size_t)(THIN_TH + 2) * (THIN_TW + 2) * (d->C0 * 2 + 16);
    const long long tiles = (long long)d->N * ((d->IH + THIN_TH - 1) / THIN_TH) * ((d->IW + THIN_TW - 1) / THIN_TW);
    if (lds > 64 * 1024 || tiles > 0x7fffffffll) return 0;
    if (d->C0 == 16) hipLaunchKernelGGL(conv_thin_out_mfma_kernel<1>, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    else if (d->C0 == 32) hipLaunchKernelGGL(conv_thin_out_mfma_kernel<2>, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    else if (d->C0 == 64) hipLaunchKernelGGL(conv_thin_out_mfma_kernel<4>, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(conv_thin_out_kernel, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    return ddpm_check_launch("conv_thin") == DDPM_OK ? 1 : -1;
}
