// Whole-image convolution tiles with the GroupNorm of the conv -> GroupNorm -> act unit (constituent_blocks.py:65-74,
// 166-188 of the reference) finished inside the convolution's own launch.
//
// GroupNorm statistics are per (image, channel group): a workgroup that owns WHOLE images and whole groups needs nobody
// else's numbers.  At the UNet's resolutions an image is small - 2x2 ... 32x32 positions of 32 ... 256 channels - so a
// workgroup takes NB whole images (all output-parity classes of a transposed / strided data-gradient form included) and BN
// output channels, and
//
//   forward    conv (bf16 MFMA, fp32 accumulate) -> raw result + per-(image, group) statistics -> act(GroupNorm(raw)) (+ residual)
//   backward   dY -> GroupNorm / activation backward in the loader (group sums complete inside the workgroup; writes the raw
//              gradient the weight-gradient launch reads and the per-image d gamma, d beta, d bias rows) -> data-gradient conv
//
// run as ONE launch each, where the general kernel needs convolution + normalisation launches with a round trip through HBM
// between them.  The input patches of the tile's images (with halo) and all weight rows of the tile's output channels are
// staged into LDS once (one barrier), every tap then reads its operand fragments at a shifted pixel offset; the result
// tile goes back through LDS so that statistics are taken over the bf16-rounded values (what a separate normalisation
// kernel would read) and global stores are 16-byte NHWC vectors.  Geometry: the tap-list form of conv_geom.h with
// TH x TW = the whole position grid of an image.
#include "conv_geom.h"
#include "ddpm_hip.h"

#include <stdlib.h>

// Diagnostic build only (-DIMG_STAMPS): lane 0 of wave 0 of every workgroup records the constant-rate clock at phase boundaries.
#ifdef IMG_STAMPS
__device__ unsigned long long* img_dbg = nullptr;
extern "C" int ddpm_img_set_dbg(unsigned long long* p) { return hipMemcpyToSymbol(HIP_SYMBOL(img_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -2; }
#define IMG_STAMP(i)                                                                                                \
    do {                                                                                                            \
        if (img_dbg && threadIdx.x == 0 && blockIdx.y == 0) img_dbg[(unsigned long long)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define IMG_STAMP(i)
#endif

struct ImgGeom {               // the scalars of ConvGeom this kernel needs (tile = NB whole images)
    int N, IH, IW, OH, OW, SI, SO, dmin_h, dmin_w, PH, PW, NB, ncls;
    unsigned pw_magic, phw_magic;
};

struct ImgArgs {
    ImgGeom g;
    // small tables, copied to LDS when the kernel starts: weight slice and patch byte offset of every tap, first tap of
    // every class, output offset of every class (off_h | off_w << 8)
    int tap_slice[CONV_MAX_TAPS], tap_poff[CONV_MAX_TAPS], cls_tap_begin[CONV_MAX_CLS + 1], cls_off[CONV_MAX_CLS];
    const bf16* in0; const bf16* in1; int C0, C1; const float* in_add;
    const bf16* w; const float* bias; int w_rows, Cin;
    bf16* out; const bf16* res; float* sum_out; int Cout, out_cs, act;
    const float* gn_gamma; const float* gn_beta; bf16* gn_out; float* gn_stats; const bf16* gn_res; int gn_groups, gn_act; float gn_eps;
    const float* gn_act_beta;        // swish / hardswish: the activation's learnable scalar (else null)
    const bf16* gnb_raw; const float* gnb_gamma; const float* gnb_beta; const float* gnb_stats; float* gnb_part; bf16* gnb_draw;
    int gnb_groups, gnb_act;
    int p_shift, gw_shift;         // log2 of positions per image (per class) and of the position grid's width
    int rows_cls;                  // NB * positions per image: rows of one class in the tile (a multiple of 32)
    int AS, WS;                    // LDS row strides in bytes: one patch pixel (Cin channels), one weight row
    int ntaps;                     // taps of all classes
    int w_off, red_off, tab_off;   // LDS byte offsets: weight rows, the loader's reduction tables, the small tables
    unsigned vpr_shift;            // log2(Cin / 8)
    int rc_shift;                  // log2(rows_cls)
    int iw_shift, ihw_shift;       // log2(IW), log2(IH * IW) (loader with the GroupNorm backward: powers of two there)
    int gn_cpg_shift, gnb_cpg_shift;   // log2(channels per group) of the fused GroupNorm / of its backward
};
enum { IMG_FWD_GN = 0, IMG_BWD_GN = 1, IMG_PLAIN = 2 };   // what a launch fuses: the unit's GroupNorm, its backward, nothing
enum { TAB_SLICE = 0, TAB_POFF = 16, TAB_CLS_TAP = 32, TAB_CLS_OFF = 40, TAB_INTS = 48 };

// arr[idx] for a table that lives in the kernel arguments: a chain of compares over compile-time indices (indexing such a
// table with a per-lane value would move the whole argument block to scratch memory)
template <int N>
__device__ __forceinline__ int table_pick(const int (&arr)[N], int idx) {
    int v = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) v = idx == i ? arr[i] : v;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// loader, plain form: patch pixels of the tile's images -> LDS, with the channel concat (decoder skip connection) and the
// broadcast time-embedding add fused
template <int T>
__device__ __forceinline__ void img_stage_plain(const ImgArgs& a, char* s_patch, int n0, int tid) {
    const ImgGeom& g = a.g;
    const int vpr = a.Cin >> 3, php = g.PH * g.PW;
    const int items = g.NB * php << a.vpr_shift;
#pragma unroll 4
    for (int idx = tid; idx < items; idx += T) {
        const int pix = idx >> a.vpr_shift, vec = idx & (vpr - 1);
        const int il = div_fast(pix, g.phw_magic), r = pix - il * php;
        const int py = div_fast(r, g.pw_magic), px = r - py * g.PW;
        const int n = n0 + il, iy = py + g.dmin_h, ix = px + g.dmin_w;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
        if (n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
            const long long p = ((long long)n * g.IH + iy) * g.IW + ix;
            const int c = vec * 8;
            if (c < a.C0) {
                v = *(const bf16x8*)(a.in0 + p * a.C0 + c);
                if (a.in_add) {
                    const float* ad = a.in_add + (long long)n * a.C0 + c;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + ad[j]);
                }
            } else {
                v = *(const bf16x8*)(a.in1 + p * a.C1 + (c - a.C0));
            }
        }
        *(bf16x8*)(s_patch + pix * a.AS + vec * 16) = v;
    }
}

// Sums of 16 per-lane values over the lanes that share a channel vector (lanes with equal lane % vpr), by folding: at every
// exchange a lane keeps one half of its values and hands the other half over, so the whole reduction costs 8 + 4 + 2 + 1
// cross-lane moves instead of 16 per exchange.  Afterwards a lane holds 16 >> STEPS sums, of consecutive value indices from
// `base` on, and stores them into the slot's row: value index 0..7 = first sum of channel idx, 8..15 = second sum of channel
// idx - 8.  All register indices are compile-time constants (a run-time index into a register array turns into a chain of
// compares and selects per access).
template <int STEPS>
__device__ __forceinline__ void gnb_flush_steps(float (&v)[16], float* dst, int lane) {
    int base = 0;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
        const int o = 32 >> st, n = 8 >> st;          // exchange distance; values kept after this exchange
        const bool hi = (lane & o) != 0;
        if (hi) base += n;
        // bitwise blends, not `hi ? v[i] : v[i + n]`: the compiler turns a select between two array elements into an indexed
        // access, and an indexed access to a register array into sixteen compares and selects
        const unsigned m = hi ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const unsigned lo_b = __float_as_uint(v[i]), hi_b = __float_as_uint(v[i + n]);
            const float send = __uint_as_float((lo_b & m) | (hi_b & ~m));
            const float keep = __uint_as_float((hi_b & m) | (lo_b & ~m));
            v[i] = keep + __shfl_xor(send, o, 64);
        }
    }
#pragma unroll
    for (int i = 0; i < (16 >> STEPS); ++i) {
        const int idx = base + i;
        dst[(idx & 7) * 2 + (idx >> 3)] = v[i];
    }
}
__device__ __forceinline__ void gnb_flush(float (&acc)[16], float* s_part, int slot, int c_shift, int vec, int lane, int vpr) {
    float* dst = s_part + ((slot << c_shift) + vec * 8) * 2;
    switch (vpr) {
        case 4: gnb_flush_steps<4>(acc, dst, lane); break;
        case 8: gnb_flush_steps<3>(acc, dst, lane); break;
        case 16: gnb_flush_steps<2>(acc, dst, lane); break;
        case 32: gnb_flush_steps<1>(acc, dst, lane); break;
        default: gnb_flush_steps<0>(acc, dst, lane); break;
    }
}

// loader, GroupNorm-backward form (see the file header).  item = (image of the tile, pixel, 8-channel vector) of dY, in that
// order; thread t owns items t, t + T, ... and loads GNB_KI of them per round of global loads (one round per pass for all
// but the largest images; the second pass reads them again: cache hits).  A lane's vector index is the same for all its
// items.  Sums over pixels: per lane over its items (one image per tile) or per item (several images per tile: 64
// consecutive items lie in one image) -> folded over the lanes that share a vector -> per-slot rows in LDS (slot = wave, or
// item / 64) -> the slots of an image added in a fixed order.
//   pass 1: a1 = sum g, a2 = sum g xh    (g = dy act'(xh gamma + beta), xh = (x - mean) rstd)  -> d beta, d gamma, S1, S2
//   pass 2: dd = rstd (gamma g - (S1 + xh S2)) -> LDS patch, global raw gradient; sum dd -> bias gradient
// s_red: [NB][G][2] mean, rstd | [NB][G][2] S1, S2 | [NB][C][2] sums | [slots][C][2] per-slot sums.
#define GNB_KI 2
template <int T, int ACT>
__device__ __forceinline__ void img_stage_gnb(const ImgArgs& a, char* s_patch, float* s_red, int n0, int tid, bool write_global) {
    const ImgGeom& g = a.g;
    constexpr int NW = T / 64;
    const int C = a.Cin, G = a.gnb_groups, cpg = 1 << a.gnb_cpg_shift, vpr = C >> 3;
    const int lane = tid & 63, wave = tid >> 6;
    const int c_shift = a.vpr_shift + 3, g_shift = c_shift - a.gnb_cpg_shift;
    const int ipi_shift = a.ihw_shift + a.vpr_shift;                 // log2(items per image)
    const int items_total = g.NB << ipi_shift;
    const bool one = g.NB == 1;
    const int spi = one ? NW : 1 << (ipi_shift - 6);                // slots per image
    float* s_mr = s_red;
    float* s_S = s_mr + g.NB * G * 2;
    float* s_sum = s_S + g.NB * G * 2;
    float* s_part = s_sum + g.NB * C * 2;
    // the tile's statistics -> LDS
    for (int idx = tid; idx < g.NB * G * 2; idx += T) {
        const int il = idx >> (g_shift + 1);
        s_mr[idx] = n0 + il < g.N ? a.gnb_stats[(long long)n0 * G * 2 + idx] : 0.f;
    }
    const int vec = lane & (vpr - 1);
    float ga[8], be[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = a.gnb_gamma[vec * 8 + j]; be[j] = a.gnb_beta[vec * 8 + j]; }
    // halo (and the patches of images past the end of the batch) read as zero
    {
        const int php = g.PH * g.PW, items = g.NB * php << a.vpr_shift;
        for (int idx = tid; idx < items; idx += T) {
            const int pix = idx >> a.vpr_shift, v8 = idx & (vpr - 1);
            const int il = div_fast(pix, g.phw_magic), r = pix - il * php;
            const int py = div_fast(r, g.pw_magic), px = r - py * g.PW;
            const int iy = py + g.dmin_h, ix = px + g.dmin_w;
            if (!(n0 + il < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW)) {
                bf16x8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
                *(bf16x8*)(s_patch + pix * a.AS + v8 * 16) = z;
            }
        }
    }
    __syncthreads();
    IMG_STAMP(8);
#pragma unroll 1
    for (int pass = 1; pass <= 2; ++pass) {
        float mean[8], rstd[8], S1[8], S2[8];
        float tot[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) tot[i] = 0.f;
        if (one) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gi = (vec * 8 + j) >> a.gnb_cpg_shift;
                mean[j] = s_mr[gi * 2]; rstd[j] = s_mr[gi * 2 + 1];
                S1[j] = pass == 2 ? s_S[gi * 2] : 0.f; S2[j] = pass == 2 ? s_S[gi * 2 + 1] : 0.f;
            }
        }
        const int act_c = ACT >= 0 ? ACT : a.gnb_act;
#pragma unroll 1
        for (int base = 0; base < items_total; base += GNB_KI * T) {
            bf16x8 xv[GNB_KI], dv[GNB_KI];
#pragma unroll
            for (int k = 0; k < GNB_KI; ++k) {
                const int item = base + tid + k * T;
#pragma unroll
                for (int j = 0; j < 8; ++j) { xv[k][j] = (bf16)0.f; dv[k][j] = (bf16)0.f; }
                if (item < items_total && n0 + (item >> ipi_shift) < g.N) {
                    const long long e = (((long long)n0 << a.ihw_shift) + (item >> a.vpr_shift)) * C + vec * 8;
                    xv[k] = *(const bf16x8*)(a.gnb_raw + e);
                    dv[k] = *(const bf16x8*)(a.in0 + e);
                }
            }
#pragma unroll
            for (int k = 0; k < GNB_KI; ++k) {
                const int item = base + tid + k * T;
                if (item >= items_total) continue;               // wave-uniform: items_total is a multiple of 64
                const int il = item >> ipi_shift;
                if (!one) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int gi = (il << g_shift) + ((vec * 8 + j) >> a.gnb_cpg_shift);
                        mean[j] = s_mr[gi * 2]; rstd[j] = s_mr[gi * 2 + 1];
                        S1[j] = pass == 2 ? s_S[gi * 2] : 0.f; S2[j] = pass == 2 ? s_S[gi * 2 + 1] : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) tot[i] = 0.f;
                }
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = ((float)xv[k][j] - mean[j]) * rstd[j];
                    const float gv = (float)dv[k][j] * act_bwd(xh * ga[j] + be[j], act_c);
                    if (pass == 1) {
                        tot[j] += gv; tot[8 + j] += gv * xh;
                    } else {
                        const float dd = rstd[j] * (ga[j] * gv - (S1[j] + xh * S2[j]));
                        tot[j] += dd;
                        o[j] = (bf16)dd;
                    }
                }
                if (pass == 2) {
                    const int pix = (item >> a.vpr_shift) & ((1 << a.ihw_shift) - 1);
                    const int iy = pix >> a.iw_shift, ix = pix & (g.IW - 1);
                    const int pp = (il * g.PH + (iy - g.dmin_h)) * g.PW + (ix - g.dmin_w);
                    *(bf16x8*)(s_patch + pp * a.AS + vec * 16) = o;
                    if (write_global && n0 + il < g.N)
                        *(bf16x8*)(a.gnb_draw + (((long long)n0 << a.ihw_shift) + (item >> a.vpr_shift)) * C + vec * 8) = o;
                }
                if (!one) gnb_flush(tot, s_part, item >> 6, c_shift, vec, lane, vpr);
            }
        }
        if (one) gnb_flush(tot, s_part, wave, c_shift, vec, lane, vpr);
        IMG_STAMP(pass == 1 ? 9 : 12);
        __syncthreads();
        // the slots of an image, added in a fixed order
        for (int idx = tid; idx < g.NB * C; idx += T) {
            const int il = idx >> c_shift, c = idx & (C - 1);
            float u = 0.f, v = 0.f;
            for (int sl = il * spi; sl < (il + 1) * spi; ++sl) { u += s_part[((sl << c_shift) + c) * 2]; v += s_part[((sl << c_shift) + c) * 2 + 1]; }
            const long long n = n0 + il;
            if (pass == 1) {
                s_sum[idx * 2] = u; s_sum[idx * 2 + 1] = v;
                if (n < g.N && write_global) {
                    a.gnb_part[(n * 3 + 0) * C + c] = v;          // d gamma = sum g xh
                    a.gnb_part[(n * 3 + 1) * C + c] = u;          // d beta  = sum g
                }
            } else if (n < g.N && write_global) {
                a.gnb_part[(n * 3 + 2) * C + c] = u;              // bias gradient of the unit's convolution = sum of the raw gradient
            }
        }
        IMG_STAMP(pass == 1 ? 10 : 13);
        if (pass == 1) {
            __syncthreads();
            const float im = 1.0f / (float)(cpg << a.ihw_shift);
            for (int idx = tid; idx < g.NB * G; idx += T) {
                const int il = idx >> g_shift, grp = idx & (G - 1);
                float u = 0.f, v = 0.f;
                for (int c = grp * cpg; c < (grp + 1) * cpg; ++c) {
                    const float gm = a.gnb_gamma[c];
                    u += gm * s_sum[((il << c_shift) + c) * 2]; v += gm * s_sum[((il << c_shift) + c) * 2 + 1];
                }
                s_S[idx * 2] = u * im; s_S[idx * 2 + 1] = v * im;
            }
            __syncthreads();
            IMG_STAMP(11);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// WM x WN waves; a wave owns MT x NT blocks of 32 rows x 32 output channels.  rows of the tile = [class][image][position].
template <int WM, int WN, int MT, int NT, int T, int MODE>
__global__ __launch_bounds__(T) void conv_img_kernel(const ImgArgs a) {
    constexpr int ROWS = WM * MT * 32, BN = WN * NT * 32;
    static_assert(T >= WM * WN * 64 && T % (BN / 8) == 0, "more tiles than waves");
    constexpr int OS = BN * 2 + 16;        // result tile row stride (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ImgGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform, and the compiler must know it: everything
                                                                  // derived from it (class, tap range, loop bounds) stays scalar
    const bool mm = wid < WM * WN;              // waves beyond the tile grid only help the loader and the epilogue
    const int wm = mm ? wid % WM : 0, wn = mm ? wid / WM : 0;
    const int n0 = blockIdx.x * g.NB, co0 = blockIdx.y * BN;
    char* s_patch = smem;
    char* s_w = smem + a.w_off;
    int* s_tab = (int*)(smem + a.tab_off);
    IMG_STAMP(0);
    if (tid < CONV_MAX_TAPS) {
        s_tab[TAB_SLICE + tid] = table_pick(a.tap_slice, tid);
        s_tab[TAB_POFF + tid] = table_pick(a.tap_poff, tid);
        if (tid <= CONV_MAX_CLS) s_tab[TAB_CLS_TAP + tid] = table_pick(a.cls_tap_begin, tid);
        if (tid < CONV_MAX_CLS) s_tab[TAB_CLS_OFF + tid] = table_pick(a.cls_off, tid);
    }
    __syncthreads();

    // ---- staging: weight rows of this tile's output channels for every tap ([tap][BN][Cin]) and the input patches -> LDS
    if (MODE == IMG_BWD_GN) {
        const int vpr = a.Cin >> 3;
        const int items = a.ntaps * BN << a.vpr_shift;
#pragma unroll 8
        for (int idx = tid; idx < items; idx += T) {
            const int row = idx >> a.vpr_shift, vec = idx & (vpr - 1);
            const int j = row / BN, c = row - j * BN;
            const bf16x8 v = *(const bf16x8*)(a.w + ((long long)s_tab[TAB_SLICE + j] * a.w_rows + co0 + c) * a.Cin + vec * 8);
            *(bf16x8*)(s_w + row * a.WS + vec * 16) = v;
        }
        IMG_STAMP(1);
        float* s_red = (float*)(smem + a.red_off);
        if (a.gnb_act == ACT_GELU) img_stage_gnb<T, ACT_GELU>(a, s_patch, s_red, n0, tid, blockIdx.y == 0);
        else if (a.gnb_act == ACT_SILU) img_stage_gnb<T, ACT_SILU>(a, s_patch, s_red, n0, tid, blockIdx.y == 0);
        else img_stage_gnb<T, -1>(a, s_patch, s_red, n0, tid, blockIdx.y == 0);
    } else {
        // one item space (patch vectors first, weight vectors after them), walked eight items per thread at a time so that
        // the loads of both kinds are in flight together: one round trip to memory instead of two
        const int vpr = a.Cin >> 3, php = g.PH * g.PW;
        const int n_patch = g.NB * php << a.vpr_shift;
        const int items = n_patch + (a.ntaps * BN << a.vpr_shift);
#pragma unroll 8
        for (int idx = tid; idx < items; idx += T) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
            int dst;
            if (idx < n_patch) {
                const int pix = idx >> a.vpr_shift, vec = idx & (vpr - 1);
                const int il = div_fast(pix, g.phw_magic), r = pix - il * php;
                const int py = div_fast(r, g.pw_magic), px = r - py * g.PW;
                const int n = n0 + il, iy = py + g.dmin_h, ix = px + g.dmin_w;
                dst = pix * a.AS + vec * 16;
                if (n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) {
                    const long long p = ((long long)n * g.IH + iy) * g.IW + ix;
                    const int c = vec * 8;
                    if (c < a.C0) {
                        v = *(const bf16x8*)(a.in0 + p * a.C0 + c);
                        if (a.in_add) {
                            const float* ad = a.in_add + (long long)n * a.C0 + c;
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + ad[j]);
                        }
                    } else {
                        v = *(const bf16x8*)(a.in1 + p * a.C1 + (c - a.C0));
                    }
                }
            } else {
                const int wi = idx - n_patch;
                const int row = wi >> a.vpr_shift, vec = wi & (vpr - 1);
                const int j = row / BN, c = row - j * BN;
                v = *(const bf16x8*)(a.w + ((long long)s_tab[TAB_SLICE + j] * a.w_rows + co0 + c) * a.Cin + vec * 8);
                dst = a.w_off + row * a.WS + vec * 16;
            }
            *(bf16x8*)(smem + dst) = v;
        }
        IMG_STAMP(1);
    }
    IMG_STAMP(2);

    // ---- this wave's rows
    const int rb0 = wm * MT;                                   // first 32-row block
    const int cls = (rb0 * 32) >> a.rc_shift;                  // all blocks of a wave lie in one class (host-checked)
    const int tap0 = s_tab[TAB_CLS_TAP + cls], ntaps = s_tab[TAB_CLS_TAP + cls + 1] - tap0;
    const int my_poff = lane < ntaps ? s_tab[TAB_POFF + tap0 + lane] : 0;
    int a_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = (rb0 + mt) * 32 + (lane & 31) - cls * a.rows_cls;
        const int il = r >> a.p_shift, pos = r & ((1 << a.p_shift) - 1);
        const int py = pos >> a.gw_shift, px = pos & ((1 << a.gw_shift) - 1);
        a_base[mt] = ((il * g.PH + py * g.SI) * g.PW + px * g.SI) * a.AS + (lane >> 5) * 16;
    }
    const int b_base = (wn * NT * 32 + (lane & 31)) * a.WS + (lane >> 5) * 16;
    f32x16 acc[MT][NT];
    float bias_r[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + wn * NT * 32 + nt * 32 + (lane & 31);
        bias_r[nt] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    }
    // the epilogue's per-channel operands: a thread's 8-channel vector is the same for all its result vectors (T is a multiple
    // of BN / 8), so they are fetched here, where the latency hides behind the matrix work
    const int my_part = tid % (BN / 8);
    f32x4 gm0, gm1, bt0, bt1;
    if (MODE == IMG_FWD_GN) {
        const int co = co0 + my_part * 8;
        gm0 = *(const f32x4*)(a.gn_gamma + co); gm1 = *(const f32x4*)(a.gn_gamma + co + 4);
        bt0 = *(const f32x4*)(a.gn_beta + co); bt1 = *(const f32x4*)(a.gn_beta + co + 4);
    }
    __syncthreads();
    IMG_STAMP(3);

    // ---- MFMA over taps x k-steps of 16 channels; the operand fragments of step s + D are requested right after step s has
    //      been issued (steady-state loop without conditions, so that the waits count exactly; the last 2 D - 1 steps peeled)
    if (mm) {
        constexpr int D = MT * NT >= 4 ? 2 : 4;
        const int ks_shift = a.vpr_shift - 1, ksteps = 1 << ks_shift;       // Cin / 16
        const int total = ntaps << ks_shift;
        const int tap_stride = BN * a.WS;
        bf16x8 fa[D][MT], fb[D][NT];
        auto fetch = [&](int st, bf16x8 (&av)[MT], bf16x8 (&bv)[NT]) {
            const int j = st >> ks_shift, kk = st & (ksteps - 1);
            const int poff = __builtin_amdgcn_readlane(my_poff, j);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = *(const bf16x8*)(s_patch + a_base[mt] + poff + kk * 32);
            const char* bp = s_w + (tap0 + j) * tap_stride + b_base + kk * 32;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = *(const bf16x8*)(bp + nt * 32 * a.WS);
        };
        auto step = [&](bf16x8 (&av)[MT], bf16x8 (&bv)[NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        };
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < total) fetch(d, fa[d], fb[d]);
        int st = 0;
        for (; st + 2 * D <= total; st += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                step(fa[d], fb[d]);
                fetch(st + D + d, fa[d], fb[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (st + d < total) {
                step(fa[d], fb[d]);
                if (st + D + d < total) fetch(st + D + d, fa[d], fb[d]);
            }
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (st + D + d < total) step(fa[d], fb[d]);
    }

    IMG_STAMP(4);
    // ---- per-image sums of the result from the fp32 accumulators (gradient of a broadcast time-embedding add)
    if (MODE != IMG_FWD_GN && a.sum_out && mm) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = co0 + wn * NT * 32 + nt * 32 + (lane & 31);
            float run = 0.f;
            int run_n = -1;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (rb0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) - cls * a.rows_cls;
                    const int n = n0 + (row >> a.p_shift);
                    if (n >= g.N) continue;
                    if (n != run_n) {
                        if (run_n >= 0 && co < a.Cout) atomicAdd(a.sum_out + (long long)run_n * a.Cout + co, run);
                        run = 0.f; run_n = n;
                    }
                    run += a.act ? act_fwd(acc[mt][nt][r] + bias_r[nt], a.act) : acc[mt][nt][r] + bias_r[nt];
                }
            if (run_n >= 0 && co < a.Cout) atomicAdd(a.sum_out + (long long)run_n * a.Cout + co, run);
        }
    }

    // ---- result tile -> LDS (bf16)
    __syncthreads();                       // every wave is done with the operand tiles
    char* s_out = smem;
    float* s_st = (float*)(smem + ROWS * OS);     // [images of the tile][groups of the tile][2] mean, rstd
    float* s_stp = s_st + 2 * BN;                 // one image per tile: [WM][groups of the tile][2] per-wave sums
    const bool reg_stats = MODE == IMG_FWD_GN && g.NB == 1;
    if (mm) {
        if (MODE != IMG_FWD_GN && a.act) { // rare: a convolution-side activation (the units this kernel serves have none)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll 1
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = act_fwd(acc[mt][nt][r] + bias_r[nt], a.act) - bias_r[nt];
        }
        float st1[NT], st2[NT];                    // this lane's column sums over its rows, of the bf16 values
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) st1[nt] = st2[nt] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = wn * NT * 32 + nt * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (rb0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const bf16 hv = (bf16)(acc[mt][nt][r] + bias_r[nt]);
                    *(bf16*)(s_out + row * OS + col * 2) = hv;
                    if (MODE == IMG_FWD_GN) { const float f = (float)hv; st1[nt] += f; st2[nt] += f * f; }
                }
            }
        if (reg_stats) {
            // one image per tile: every row of the tile belongs to it, so the statistics come straight from the registers:
            // lanes of a group (and the two row halves of the wave) -> per-wave sums -> added over the row waves below
            const int cpg_ = 1 << a.gn_cpg_shift;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float u = st1[nt], v = st2[nt];
                u += __shfl_xor(u, 32, 64); v += __shfl_xor(v, 32, 64);
                for (int o = 1; o < cpg_; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
                if (lane < 32 && (lane & (cpg_ - 1)) == 0) {
                    const int grp = (wn * NT * 32 + nt * 32 + lane) >> a.gn_cpg_shift;
                    s_stp[(wm * (BN >> a.gn_cpg_shift) + grp) * 2] = u;
                    s_stp[(wm * (BN >> a.gn_cpg_shift) + grp) * 2 + 1] = v;
                }
            }
        }
    }
    __syncthreads();
    IMG_STAMP(5);

    const int P = 1 << a.p_shift;
    int cpg = 1, gpt = 0;
    if (MODE == IMG_FWD_GN) {
        // ---- statistics per (image of the tile, group of the tile) over the bf16 values of every class
        cpg = 1 << a.gn_cpg_shift; gpt = BN >> a.gn_cpg_shift;
        const int gpt_shift = __builtin_ctz(BN) - a.gn_cpg_shift;
        if (reg_stats) {
            if (tid < gpt) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { s1 += s_stp[(w * gpt + tid) * 2]; s2 += s_stp[(w * gpt + tid) * 2 + 1]; }
                const float m = 1.0f / (float)((g.ncls << a.p_shift) * cpg);
                const float mean = s1 * m;
                const float rstd = rsqrtf(fmaxf(s2 * m - mean * mean, 0.f) + a.gn_eps);
                s_st[tid * 2] = mean; s_st[tid * 2 + 1] = rstd;
                if (n0 < g.N && a.gn_stats) {
                    const int gi = (co0 >> a.gn_cpg_shift) + tid;
                    a.gn_stats[((long long)n0 * a.gn_groups + gi) * 2] = mean;
                    a.gn_stats[((long long)n0 * a.gn_groups + gi) * 2 + 1] = rstd;
                }
            }
        } else {
        const int npairs = g.NB * gpt;
        int tpp = T / npairs;                                  // threads per pair: a power of two, at most one wave
        if (tpp > 64) tpp = 64;
        if (tpp < 1) tpp = 1;
        const int cw = cpg < 8 ? cpg : 8, parts_shift = cpg > 8 ? a.gn_cpg_shift - 3 : 0;   // channels per LDS read
        const int rows_img = g.ncls << a.p_shift;
        const int tpp_shift = __builtin_ctz(tpp);
        for (int pair0 = 0; pair0 < npairs; pair0 += T >> tpp_shift) {
            const int pair = pair0 + (tid >> tpp_shift), sub = tid & (tpp - 1);
            float s1 = 0.f, s2 = 0.f;
            if (pair < npairs) {
                const int il = pair >> gpt_shift, grp = pair & (gpt - 1);
                for (int e = sub; e < rows_img << parts_shift; e += tpp) {
                    const int ri = e >> parts_shift, part = e & ((1 << parts_shift) - 1);
                    const int row = (ri >> a.p_shift) * a.rows_cls + il * P + (ri & (P - 1));
                    const char* src = s_out + row * OS + (grp * cpg + part * cw) * 2;
                    if (cw == 8) {
                        const bf16x8 v = *(const bf16x8*)src;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s1 += f; s2 += f * f; }
                    } else if (cw == 4) {
                        const bf16x4 v = *(const bf16x4*)src;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float f = (float)v[j]; s1 += f; s2 += f * f; }
                    } else {
                        for (int j = 0; j < cw; ++j) { const float f = (float)((const bf16*)src)[j]; s1 += f; s2 += f * f; }
                    }
                }
            }
            for (int o = 1; o < tpp; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (pair < npairs && sub == 0) {
                const int il = pair >> gpt_shift, grp = pair & (gpt - 1);
                const float m = 1.0f / (float)(rows_img * cpg);
                const float mean = s1 * m;
                const float rstd = rsqrtf(fmaxf(s2 * m - mean * mean, 0.f) + a.gn_eps);
                s_st[pair * 2] = mean; s_st[pair * 2 + 1] = rstd;
                const long long n = n0 + il;
                if (n < g.N && a.gn_stats) {
                    const int gi = (co0 >> a.gn_cpg_shift) + grp;
                    a.gn_stats[(n * a.gn_groups + gi) * 2] = mean;
                    a.gn_stats[(n * a.gn_groups + gi) * 2 + 1] = rstd;
                }
            }
        }
        }
        __syncthreads();
    }

    IMG_STAMP(6);
    // ---- tile -> global in 16-byte vectors: raw result (+ residual), and the normalised, activated one
    const int gpt_shift2 = MODE == IMG_FWD_GN ? __builtin_ctz(BN) - a.gn_cpg_shift : 0;
    act_dispatch(MODE == IMG_FWD_GN ? a.gn_act : ACT_NONE, [&](auto atag) {
    const int act_c = ACT_OF(atag, a.gn_act);
    const float act_b = (MODE == IMG_FWD_GN && a.gn_act_beta) ? *a.gn_act_beta : 1.f;
#pragma unroll 2
    for (int it = tid; it < ROWS * (BN / 8); it += T) {
        const int row = it / (BN / 8), part = it - row * (BN / 8);
        const int co = co0 + part * 8;
        if (co >= a.Cout) continue;
        const int c = row >> a.rc_shift, r = row & (a.rows_cls - 1);
        const int il = r >> a.p_shift, pos = r & (P - 1);
        const int py = pos >> a.gw_shift, px = pos & ((1 << a.gw_shift) - 1);
        const int n = n0 + il;
        if (n >= g.N) continue;
        const int coff = s_tab[TAB_CLS_OFF + c];
        const int oh = py * g.SO + (coff & 255), ow = px * g.SO + (coff >> 8);
        const long long pixel = ((long long)n * g.OH + oh) * g.OW + ow;
        const bf16x8 v = *(const bf16x8*)(s_out + row * OS + part * 16);
        if (MODE != IMG_FWD_GN || a.out) {
            bf16x8 o = v;
            const long long e = pixel * a.out_cs + co;
            if (MODE != IMG_FWD_GN && a.res) {
                const bf16x8 rv = *(const bf16x8*)(a.res + e);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)o[j] + (float)rv[j]);
            }
            *(bf16x8*)(a.out + e) = o;
        }
        if (MODE == IMG_FWD_GN) {
            const long long e = pixel * a.Cout + co;
            const f32x4 g0 = gm0, g1 = gm1, b0 = bt0, b1 = bt1;       // part == my_part
            bf16x8 rv;
#pragma unroll
            for (int j = 0; j < 8; ++j) rv[j] = (bf16)0.f;
            if (a.gn_res) rv = *(const bf16x8*)(a.gn_res + e);
            bf16x8 y;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pair = (il << gpt_shift2) + ((part * 8 + j) >> a.gn_cpg_shift);
                const float mean = s_st[pair * 2], rstd = s_st[pair * 2 + 1];
                const float gm = j < 4 ? g0[j & 3] : g1[j & 3], bt = j < 4 ? b0[j & 3] : b1[j & 3];
                const float sc = gm * rstd, sh = bt - mean * sc;
                y[j] = (bf16)(act_fwd((float)v[j] * sc + sh, act_c, act_b) + (float)rv[j]);
            }
            *(bf16x8*)(a.gn_out + e) = y;
        }
    }
    });
    IMG_STAMP(7);
}

// ------------------------------------------------------------------------------------------------ host
struct ImgPlan {
    ImgArgs a;
    int wm, wn, mt, nt, mode;
    size_t lds;
    dim3 grid;
};

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// tile shapes (rows, channels) -> tile grid of the waves.  Workgroups have 512 threads whatever
// the grid: the waves without a tile still load and store.
static bool img_layout(int rows, int bn, int* wm, int* wn, int* mt, int* nt) {
    static const int tab[][6] = {
        {256, 32, 8, 1, 1, 1}, {1024, 32, 8, 1, 4, 1}, {256, 64, 8, 1, 1, 2}, {128, 128, 4, 2, 1, 2}, {128, 64, 4, 2, 1, 1},
        {128, 32, 4, 1, 1, 1}, {64, 64, 2, 2, 1, 1}, {64, 128, 2, 4, 1, 1}, {64, 32, 2, 1, 1, 1}, {32, 256, 1, 8, 1, 1},
        {32, 128, 1, 4, 1, 1}, {32, 64, 1, 2, 1, 1}, {32, 32, 1, 1, 1, 1},
    };
    for (auto& t : tab)
        if (t[0] == rows && t[1] == bn) { *wm = t[2]; *wn = t[3]; *mt = t[4]; *nt = t[5]; return true; }
    return false;
}

// Fills the plan when this kernel can run the problem (fused GroupNorm forward or backward form included); false otherwise.
static bool img_plan(const ddpm_conv_desc* d, ImgPlan* p) {
    static const bool off = getenv("DDPM_NO_CONV_IMG") != nullptr;      // measurement switch: everything falls back to the other kernels
    if (!d || off) return false;
    const int kind = d->kind;
    if (kind != CONV_3X3_S1 && kind != CONV_3X3_S2 && kind != CONVT_2X2_S2 && kind != DGRAD_3X3_S2 && kind != DGRAD_CONVT_2X2) return false;
    if (d->out_mode != OUT_BF16_NHWC || d->in_mask || d->out_mask || d->out2 || d->C0_used) return false;
    if (d->N <= 0 || d->IH <= 0 || d->IW <= 0) return false;
    const int Cin = d->C0 + d->C1;
    if (!is_pow2(Cin) || Cin < 32 || Cin > 512 || d->C0 % 8 || d->C1 % 8) return false;
    if (d->CoutW % 32 || d->Cout % 8 || d->out_cs % 8) return false;
    if (kind == DGRAD_CONVT_2X2 && (d->IH % 2 || d->IW % 2)) return false;
    ImgArgs& a = p->a;
    ConvGeom g;
    if (conv_geom_init(&g, kind, d->N, d->IH, d->IW, 32) != 0) return false;
    if (!is_pow2(g.GH) || !is_pow2(g.GW)) return false;
    const int P = g.GH * g.GW;
    const int NB = P >= 32 ? 1 : 32 / P;
    const int rows = g.ncls * NB * P;
    // whole images: the tile's position grid is the image's
    g.TH = g.GH; g.TW = g.GW; g.NB = NB;
    g.PH = (g.GH - 1) * g.SI + (g.dmax_h - g.dmin_h) + 1;
    g.PW = (g.GW - 1) * g.SI + (g.dmax_w - g.dmin_w) + 1;
    g.tiles_h = g.tiles_w = 1; g.tiles_n = (d->N + NB - 1) / NB;
    g.pw_magic = div_magic(g.PW); g.phw_magic = div_magic(g.PH * g.PW);
    const int npix = NB * g.PH * g.PW;
    if (npix >= 65536) return false;
    int ntaps = g.cls_tap_begin[g.ncls];
    // fused forms
    if (d->gn_gamma || d->gn_groups > 0) {
        if (d->gn_groups <= 0 || d->Cout % d->gn_groups || d->Cout != d->CoutW || d->out_cs != d->Cout || d->res || d->act != 0 || d->sum_out) return false;
        const int cpg = d->Cout / d->gn_groups;
        if (!is_pow2(cpg) || cpg > 32) return false;
    }
    if (d->gnb_raw || d->gnb_groups > 0) {
        if (d->C1 != 0 || d->in_add || d->gnb_groups <= 0 || Cin % d->gnb_groups) return false;
        const int cpg = Cin / d->gnb_groups;
        if (!is_pow2(cpg) || cpg > 32 || ((d->IH * d->IW * (Cin / 8)) % 64) || !is_pow2(d->IW) || !is_pow2(d->IH)) return false;
        if (kind == CONV_3X3_S2 || kind == CONVT_2X2_S2) return false;      // the loader form belongs to data-gradient launches
    }
    const int AS = Cin * 2 + 16, WS = Cin * 2 + 16;
    // widest channel tile whose operands fit LDS and whose accumulators fit the register file
    int bn = 0, wm = 0, wn = 0, mt = 0, nt = 0;
    size_t lds = 0;
    for (int c = 256; c >= 32; c >>= 1) {
        if (d->CoutW % c) continue;
        if (!img_layout(rows, c, &wm, &wn, &mt, &nt)) continue;
        const size_t patch = ((size_t)npix * AS + 15) / 16 * 16, wbytes = (size_t)ntaps * c * WS;
        const int threads = 512;
        const size_t gnb_items = (size_t)NB * d->IH * d->IW * (Cin / 8);
        (void)threads;
        const size_t red = d->gnb_groups > 0 ? (size_t)(NB * d->gnb_groups * 4 + NB * Cin * 2 + (gnb_items / 64 > 8 ? gnb_items / 64 : 8) * Cin * 2) * 4 : 0;
        const size_t stage = patch + wbytes + red;
        const size_t outb = (size_t)rows * (c * 2 + 16) + (size_t)(NB > 1 ? NB : 2) * c * 2 * 4 + (size_t)16 * c * 8 + 64;
        const size_t body = ((stage > outb ? stage : outb) + 15) / 16 * 16;
        const size_t need = body + TAB_INTS * 4;
        if (need > 160 * 1024) continue;
        bn = c; lds = need;
        a.w_off = (int)patch; a.red_off = (int)(patch + wbytes); a.tab_off = (int)body;
        break;
    }
    if (!bn) return false;
    // all 32-row blocks of a wave must lie in one class
    const int rows_cls = NB * P;
    if ((mt * 32) > rows_cls ? true : (rows_cls % (mt * 32)) != 0) return false;
    a.g = ImgGeom{g.N, g.IH, g.IW, g.OH, g.OW, g.SI, g.SO, g.dmin_h, g.dmin_w, g.PH, g.PW, g.NB, g.ncls, g.pw_magic, g.phw_magic};
    for (int j = 0; j < CONV_MAX_TAPS; ++j) {
        a.tap_slice[j] = j < ntaps ? g.tap_slice[j] : 0;
        a.tap_poff[j] = j < ntaps ? (g.tap_dh[j] * g.PW + g.tap_dw[j]) * AS : 0;
    }
    for (int c = 0; c <= CONV_MAX_CLS; ++c) a.cls_tap_begin[c] = c <= g.ncls ? g.cls_tap_begin[c] : ntaps;
    for (int c = 0; c < CONV_MAX_CLS; ++c) a.cls_off[c] = c < g.ncls ? (g.cls_off_h[c] | (g.cls_off_w[c] << 8)) : 0;
    p->mode = d->gn_gamma ? IMG_FWD_GN : d->gnb_raw ? IMG_BWD_GN : IMG_PLAIN;
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.C0 = d->C0; a.C1 = d->C1; a.in_add = d->in_add;
    a.w = (const bf16*)d->w; a.bias = d->bias; a.w_rows = d->w_rows; a.Cin = Cin;
    a.out = (bf16*)d->out; a.res = (const bf16*)d->res; a.sum_out = d->sum_out; a.Cout = d->Cout; a.out_cs = d->out_cs; a.act = d->act;
    a.gn_gamma = d->gn_gamma; a.gn_beta = d->gn_beta; a.gn_out = (bf16*)d->gn_out; a.gn_stats = d->gn_stats; a.gn_res = (const bf16*)d->gn_res;
    a.gn_groups = d->gn_groups; a.gn_act = d->gn_act; a.gn_eps = d->gn_eps; a.gn_act_beta = d->gn_act_beta;
    a.gnb_raw = (const bf16*)d->gnb_raw; a.gnb_gamma = d->gnb_gamma; a.gnb_beta = d->gnb_beta; a.gnb_stats = d->gnb_stats;
    a.gnb_part = d->gnb_part; a.gnb_draw = (bf16*)d->gnb_draw; a.gnb_groups = d->gnb_groups; a.gnb_act = d->gnb_act;
    a.p_shift = ilog2_exact(P); a.gw_shift = ilog2_exact(g.GW);
    a.rows_cls = rows_cls; a.AS = AS; a.WS = WS; a.ntaps = ntaps;
    a.vpr_shift = (unsigned)ilog2_exact(Cin / 8);
    a.rc_shift = ilog2_exact(rows_cls);
    a.iw_shift = ilog2_exact(d->IW); a.ihw_shift = ilog2_exact(d->IW) + ilog2_exact(d->IH);
    a.gn_cpg_shift = d->gn_groups > 0 ? ilog2_exact(d->Cout / d->gn_groups) : 0;
    a.gnb_cpg_shift = d->gnb_groups > 0 ? ilog2_exact(Cin / d->gnb_groups) : 0;
    p->wm = wm; p->wn = wn; p->mt = mt; p->nt = nt; p->lds = lds;
    p->grid = dim3((unsigned)g.tiles_n, (unsigned)(d->CoutW / bn), 1);
    return true;
}

int ddpm_conv_img_fusable(const ddpm_conv_desc* d, int backward) {
    ImgPlan p;
    ddpm_conv_desc t = *d;
    if (backward) { if (t.gnb_groups <= 0) return 0; t.gnb_raw = (const void*)16; }
    else { if (t.gn_groups <= 0) return 0; t.gn_gamma = (const float*)16; }
    return img_plan(&t, &p) ? 1 : 0;
}

template <int WM, int WN, int MT, int NT, int MODE>
static int img_launch_mode(const ImgPlan& p, hipStream_t stream) {
    constexpr int T = 512;
    auto k = conv_img_kernel<WM, WN, MT, NT, T, MODE>;
    if (p.lds > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds) != hipSuccess)
        return ddpm_check_launch("conv_img: raise dynamic LDS limit");
    hipLaunchKernelGGL(k, p.grid, dim3(T), p.lds, stream, p.a);
    return ddpm_check_launch("conv_img");
}

template <int WM, int WN, int MT, int NT>
static int img_launch(const ImgPlan& p, hipStream_t stream) {
    if (p.mode == IMG_FWD_GN) return img_launch_mode<WM, WN, MT, NT, IMG_FWD_GN>(p, stream);
    if (p.mode == IMG_BWD_GN) return img_launch_mode<WM, WN, MT, NT, IMG_BWD_GN>(p, stream);
    return img_launch_mode<WM, WN, MT, NT, IMG_PLAIN>(p, stream);
}

// Returns 1 when the problem was launched here, 0 when the caller should use another kernel, < 0 on error.
int ddpm_conv_img_try(const ddpm_conv_desc* d, hipStream_t stream, int plain_too) {
    if (!d->gn_gamma && !d->gnb_raw && !plain_too) return 0;
    ImgPlan p;
    if (!img_plan(d, &p)) return 0;
    if (d->gn_gamma && !(d->gn_beta && d->gn_out)) { ddpm_set_error("conv: the fused GroupNorm needs gamma, beta and an output"); return -1; }
    if (d->gnb_raw && !(d->gnb_gamma && d->gnb_beta && d->gnb_stats && d->gnb_part && d->gnb_draw)) {
        ddpm_set_error("conv: the fused GroupNorm backward needs all gnb_* pointers");
        return -1;
    }
    if (!d->gn_gamma && !d->out) { ddpm_set_error("conv: no output"); return -1; }
    int rc = DDPM_EINVAL;
#define IMG_CASE(a_, b_, c_, d_) if (p.wm == a_ && p.wn == b_ && p.mt == c_ && p.nt == d_) rc = img_launch<a_, b_, c_, d_>(p, stream);
    IMG_CASE(8, 1, 1, 1) IMG_CASE(8, 1, 4, 1) IMG_CASE(8, 1, 1, 2) IMG_CASE(4, 2, 1, 2) IMG_CASE(4, 2, 1, 1) IMG_CASE(4, 1, 1, 1)
    IMG_CASE(2, 2, 1, 1) IMG_CASE(2, 4, 1, 1) IMG_CASE(2, 1, 1, 1) IMG_CASE(1, 8, 1, 1) IMG_CASE(1, 4, 1, 1) IMG_CASE(1, 2, 1, 1)
    IMG_CASE(1, 1, 1, 1)
#undef IMG_CASE
    return rc == DDPM_OK ? 1 : -1;
}
