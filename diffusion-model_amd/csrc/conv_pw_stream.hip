// Pointwise (1x1) convolutions of LARGE images as a streaming kernel for gfx950: the expansion and projection of the
// inverted-residual block (conv_block.py:57-76,98-118 of the reference) on 64x64 / 128x128 images at sampling batch sizes,
// where K is 32 .. 128 input channels: a few matrix instructions per kilobyte moved, so the launch is a copy with a small
// GEMM in the middle and what counts is bytes in flight, not operand reuse.
//
//   * the whole weight matrix (at most 8192 elements) lives in registers as MFMA operands for the life of a wave;
//   * a wave owns 32 consecutive positions at a time and reads them from global memory directly in operand layout (one
//     16-byte vector per lane and 16 input channels), the next 32 positions requested before the current ones are used;
//   * the product is computed transposed (rows = output channels, columns = positions), so a lane ends up with 4 consecutive
//     output channels of one position; a wave-private LDS tile turns that into whole 16-byte NHWC vectors in row order, and
//     the global stores of a wave are contiguous kilobytes;
//   * no workgroup barrier inside the loop: the four waves of a workgroup only share the bias / affine tables and meet again
//     for the GroupNorm sums at the end.
//
// Fused in: bias, the normalise-on-load of the input (per image and channel scale / shift from the producer's partial sums,
// then none / gelu / silu), the input's dropout mask, and the per-tile GroupNorm sums of the rounded result for the
// normalisation that follows; or (EPI, the data-gradient of the block's expansion) the dropout mask of the result and a tensor
// added last, with the arithmetic of the tiled kernel's epilogue (mask and sum on the bf16-rounded product).
#include "conv_geom.h"
#include "ddpm_hip.h"

namespace {

struct PwsArgs {
    const bf16* x; const bf16* w; const float* bias; bf16* out;
    const float* in_gn;              // [N][CIN][2] (scale, shift), or null
    const unsigned char* in_mask;    // keep bits of the input, one byte per 8 consecutive elements, or null
    float in_mask_scale;
    float* gn_part; int gn_G;        // [N][gn_G][PT][2], or null
    const bf16* res;                 // EPI: tensor of out's layout added last, or null
    const unsigned char* out_mask;   // EPI: keep bits of the result (applied before res), one byte per 8 consecutive elements, or null
    float out_mask_scale;
    int HW, TP, PT, out_cs;          // positions per image, per workgroup; workgroups per image; channel stride of out
};

// LDS traffic inside one wave: the hardware executes a wave's LDS instructions in order, so a write followed by a read of
// another lane's data needs no barrier, only a compiler fence that keeps the two in program order.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// PREACT: -1 = the input is used as it is; otherwise the activation code of the normalise-on-load (none / gelu / silu)
// EPI: the result is masked and / or added to a.res on its way out
template <int CIN, int COUT, int PREACT, bool EPI = false>
__global__ __launch_bounds__(256, 2) void pw_stream_kernel(const PwsArgs a) {
    constexpr int KS = CIN / 16, NT = COUT / 32, PARTS = COUT / 8;
    constexpr int OS = COUT * 2 + 16;             // row stride of the wave's output tile in LDS (bytes)
    constexpr int WAVE_LDS = 32 * OS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = (float*)(smem + 4 * WAVE_LDS);
    float* s_aff = s_bias + COUT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / a.PT, tile = blockIdx.x - n * a.PT;

    if (tid < COUT) s_bias[tid] = a.bias ? a.bias[tid] : 0.f;
    if constexpr (PREACT >= 0) {
        for (int i = tid; i < CIN * 2; i += 256) s_aff[i] = a.in_gn[(long long)n * CIN * 2 + i];
    }
    // the weights as the A operand (rows = output channels): lane = (row & 31), 8 input channels at 16 * kk + 8 * (lane >> 5)
    bf16x8 wr[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            wr[nt][kk] = *(const bf16x8*)(a.w + (nt * 32 + (lane & 31)) * CIN + kk * 16 + (lane >> 5) * 8);
    __syncthreads();

    const long long pos0 = (long long)n * a.HW + (long long)tile * a.TP;
    const bf16* xb = a.x + pos0 * CIN;            // workgroup-uniform bases, 32-bit byte offsets per lane from here on
    const unsigned char* mb = a.in_mask ? a.in_mask + pos0 * (CIN / 8) : nullptr;
    bf16* ob = a.out + pos0 * a.out_cs;
    const int nsub = a.TP >> 5;
    const unsigned foff = ((lane & 31) * CIN + (lane >> 5) * 8) * 2;
    char* stg = smem + wave * WAVE_LDS;
    const int part = lane % PARTS, row_in = lane / PARTS;       // this lane's 16-byte column and first row of the output tile
    constexpr int RSTEP = 64 / PARTS;                           // rows covered by one pass of the wave
    constexpr int NPASS = 32 / RSTEP;
    const bf16* rb = EPI && a.res ? a.res + pos0 * a.out_cs : nullptr;
    const unsigned char* omb = EPI && a.out_mask ? a.out_mask + pos0 * a.out_cs / 8 : nullptr;      // (out_cs is a multiple of 8)
    // what the epilogue adds and masks with, per pass of the wave over its tile: requested one sub-tile ahead, like the input
    auto load_epi = [&](int s, bf16x8 (&rv)[NPASS], unsigned (&om)[NPASS]) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const unsigned off = (unsigned)((s * 32 + i * RSTEP + row_in) * a.out_cs + part * 8);
            if (rb) rv[i] = ld_at<bf16x8>(rb, off * 2);
            if (omb) om[i] = ld_at<unsigned char>(omb, off >> 3);
        }
    };

    // (the keep bits of a vector sit at its element offset / 8: one byte per lane and 16 input channels, fetched with the data)
    auto load = [&](int s, bf16x8 (&v)[KS], unsigned (&m)[KS]) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) v[kk] = ld_at<bf16x8>(xb, (unsigned)s * (32 * CIN * 2) + foff + kk * 32);
        if (mb) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) m[kk] = ld_at<unsigned char>(mb, (unsigned)s * (32 * CIN / 8) + (foff >> 4) + kk * 2);
        }
    };
    bf16x8 xr[KS];
    unsigned mr[KS] = {};
    load(wave, xr, mr);
    bf16x8 rr[NPASS] = {};
    unsigned orr[NPASS] = {};
    if constexpr (EPI) load_epi(wave, rr, orr);
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_ss[4] = {0.f, 0.f, 0.f, 0.f};

    for (int s = wave; s < nsub; s += 4) {
        bf16x8 xn[KS];
        unsigned mn[KS] = {};
        load(s + 4 < nsub ? s + 4 : s, xn, mn);   // (the last trip re-reads its own positions: one cached kilobyte)
        bf16x8 rn[NPASS] = {};
        unsigned on[NPASS] = {};
        if constexpr (EPI) load_epi(s + 4 < nsub ? s + 4 : s, rn, on);
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 f = xr[kk];
            if constexpr (PREACT >= 0) {
                // normalise + activate what the producer left raw, rounded to bf16 as a GroupNorm launch would have stored it
                const f32x4* af = (const f32x4*)(s_aff + (kk * 16 + (lane >> 5) * 8) * 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 t = af[k];
                    f[2 * k] = (bf16)act_fwd((float)f[2 * k] * t[0] + t[1], PREACT);
                    f[2 * k + 1] = (bf16)act_fwd((float)f[2 * k + 1] * t[2] + t[3], PREACT);
                }
            }
            if (mb) {
                const unsigned m = mr[kk];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = (m >> j) & 1 ? (bf16)((float)f[j] * a.in_mask_scale) : (bf16)0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[nt][kk], f, acc[nt], 0, 0, 0);
        }
        // accumulator element r of a lane = output channel (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) of position (lane & 31)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = nt * 32 + q * 8 + 4 * (lane >> 5);
                const f32x4 b = *(const f32x4*)(s_bias + c0);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)(acc[nt][4 * q + j] + b[j]);
                *(bf16x4*)(stg + (lane & 31) * OS + c0 * 2) = o;
            }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 32 / RSTEP; ++i) {
            const int row = i * RSTEP + row_in;
            bf16x8 v = *(const bf16x8*)(stg + row * OS + part * 16);
            pair_sums(v, st_s, st_ss);
            if constexpr (EPI) {
                if (omb) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (orr[i] >> j) & 1 ? (bf16)((float)v[j] * a.out_mask_scale) : (bf16)0.f;
                }
                if (rb) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)rr[i][j]);
                }
            }
            st_at(ob, (unsigned)((s * 32 + row) * a.out_cs + part * 8) * 2, v);
        }
        wave_lds_fence();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) { xr[kk] = xn[kk]; mr[kk] = mn[kk]; }
        if constexpr (EPI) {
#pragma unroll
            for (int i = 0; i < NPASS; ++i) { rr[i] = rn[i]; orr[i] = on[i]; }
        }
    }

    if (a.gn_part) {
        // lanes that share a 16-byte column add up by shuffles, the four waves' column sums meet in LDS, and one thread per
        // group adds its channel pairs in a fixed order and stores the workgroup's pair
#pragma unroll
        for (int o = PARTS; o < 64; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { st_s[q] += __shfl_xor(st_s[q], o); st_ss[q] += __shfl_xor(st_ss[q], o); }
        }
        __syncthreads();
        float* red = (float*)smem;                 // [wave][column][4 sums, 4 sums of squares]
        if (lane < PARTS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { red[(wave * PARTS + lane) * 8 + q] = st_s[q]; red[(wave * PARTS + lane) * 8 + 4 + q] = st_ss[q]; }
        }
        __syncthreads();
        const int cpg = COUT / a.gn_G;
        if (tid < a.gn_G) {
            float S = 0.f, SS = 0.f;
            for (int c2 = tid * cpg / 2; c2 < (tid + 1) * cpg / 2; ++c2) {
                const int col = c2 >> 2, q = c2 & 3;
#pragma unroll
                for (int w = 0; w < 4; ++w) { S += red[(w * PARTS + col) * 8 + q]; SS += red[(w * PARTS + col) * 8 + 4 + q]; }
            }
            float* o = a.gn_part + (((long long)n * a.gn_G + tid) * a.PT + tile) * 2;
            o[0] = S; o[1] = SS;
        }
    }
}

template <int CIN, int COUT, int PREACT, bool EPI = false>
int pws_launch(const PwsArgs& a, int N, hipStream_t stream) {
    constexpr size_t lds = 4 * 32 * (COUT * 2 + 16) + COUT * 4 + CIN * 8;
    hipLaunchKernelGGL((pw_stream_kernel<CIN, COUT, PREACT, EPI>), dim3((unsigned)(N * a.PT)), dim3(256), lds, stream, a);
    return ddpm_check_launch("conv_pw_stream");
}

template <int CIN, int COUT>
int pws_launch_act(const PwsArgs& a, int N, int act, hipStream_t stream) {
    if (a.res || a.out_mask) return pws_launch<CIN, COUT, -1, true>(a, N, stream);      // (never together with in_gn: pws_plan)
    if (!a.in_gn) return pws_launch<CIN, COUT, -1>(a, N, stream);
    if (act == ACT_GELU) return pws_launch<CIN, COUT, ACT_GELU>(a, N, stream);
    if (act == ACT_SILU) return pws_launch<CIN, COUT, ACT_SILU>(a, N, stream);
    return pws_launch<CIN, COUT, ACT_NONE>(a, N, stream);
}

thread_local int g_pws_mode = 0;              // 0: by problem size; 1: wherever the shape allows; -1: never

// Positions per workgroup for a problem this kernel takes, 0 for every other problem.
int pws_plan(const ddpm_conv_desc* d) {
    if (g_pws_mode < 0) return 0;
    if (d->kind != CONV_1X1 || d->C1 || d->in1 || d->in_add || d->act != ACT_NONE) return 0;
    if ((d->out_mask || d->res) && (d->in_affine || d->gn_part_groups > 0)) return 0;
    if (d->out_mode != OUT_BF16_NHWC || !d->out || d->out2 || d->sum_out || d->gn_gamma || d->gnb_raw || d->C0_used) return 0;
    if (d->Cout != d->CoutW || d->out_cs % 8 || d->out_cs < d->Cout) return 0;
    const int ci = d->C0, co = d->Cout;
    if ((ci != 32 && ci != 64 && ci != 128) || (co != 32 && co != 64 && co != 128) || ci * co > 8192) return 0;
    if (d->in_affine && d->in_affine_act != ACT_NONE && d->in_affine_act != ACT_GELU && d->in_affine_act != ACT_SILU) return 0;
    if (d->gn_part_groups > 0) {
        if (co % d->gn_part_groups) return 0;
        const int cpg = co / d->gn_part_groups;
        if (cpg < 2 || (cpg & (cpg - 1))) return 0;
    }
    const long long hw = (long long)d->IH * d->IW, rows = hw * d->N;
    if (d->N <= 0 || hw % 128 || rows * (ci > d->out_cs ? ci : d->out_cs) >= (1ll << 31)) return 0;
    // by size: from one 128-position workgroup per CU on it is at least as fast as the tiled kernels (measured), twice as fast
    // from a million positions on
    if (g_pws_mode == 0 && rows < 32 * 1024) return 0;
    for (int tp = 1024; tp > 128; tp >>= 1)
        if (hw % tp == 0 && rows / tp >= 2048) return tp;
    return 128;
}

}  // namespace

// Test / measurement hook (no counterpart in the reference): 1 = take the streaming pointwise kernel wherever the shape allows
// (small test problems would otherwise never reach it), -1 = never, 0 = by problem size.  Per calling thread; returns the
// previous setting.
extern "C" int ddpm_conv_pw_stream_mode(int mode) {
    const int prev = g_pws_mode;
    if (mode >= -1 && mode <= 1) g_pws_mode = mode;
    return prev;
}

// Workgroups per image (= partial-sum slots per image and group) when this kernel takes the problem, else 0.
int ddpm_conv_pw_stream_tiles(const ddpm_conv_desc* d) {
    const int tp = pws_plan(d);
    return tp ? d->IH * d->IW / tp : 0;
}

// Returns 1 when the problem was launched here, 0 when the caller should use another kernel, < 0 on error.
int ddpm_conv_pw_stream_try(const ddpm_conv_desc* d, hipStream_t stream) {
    const int tp = pws_plan(d);
    if (!tp) return 0;
    PwsArgs a;
    a.x = (const bf16*)d->in0; a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (bf16*)d->out;
    a.in_gn = d->in_affine; a.in_mask = d->in_mask; a.in_mask_scale = d->in_mask_scale;
    a.gn_part = d->gn_part; a.gn_G = d->gn_part ? d->gn_part_groups : 0;
    a.res = (const bf16*)d->res; a.out_mask = d->out_mask; a.out_mask_scale = d->out_mask_scale;
    a.HW = d->IH * d->IW; a.TP = tp; a.PT = a.HW / tp; a.out_cs = d->out_cs;
    int rc = DDPM_EINVAL;
#define PWS_CASE(CI, CO) if (d->C0 == CI && d->Cout == CO) rc = pws_launch_act<CI, CO>(a, d->N, d->in_affine_act, stream);
    PWS_CASE(32, 32) PWS_CASE(32, 64) PWS_CASE(32, 128) PWS_CASE(64, 32) PWS_CASE(64, 64) PWS_CASE(64, 128) PWS_CASE(128, 32) PWS_CASE(128, 64)
#undef PWS_CASE
    return rc == DDPM_OK ? 1 : -1;
}
