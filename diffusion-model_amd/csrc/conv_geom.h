// Geometry shared by the implicit-GEMM convolution kernels (forward / data-gradient in conv_mfma.hip,
// weight-gradient in conv_wgrad.hip).
//
// Every dense convolution of the UNet - 3x3 stride 1/2, 1x1, transposed 2x2 stride 2, and the
// data-gradients of all of them - is expressed as a "tap-list" convolution over a grid of POSITIONS:
//
//     out[n, py*SO + off_h, px*SO + off_w, co] = sum over taps j, ci of
//                 in[n, py*SI + dmin_h + dh_j, px*SI + dmin_w + dw_j, ci] * W[slice_j][co][ci]
//
// with out-of-range input pixels reading as zero.  A launch may hold up to four CLASSES (blockIdx.z),
// each with its own tap sub-list and output offset; that is how stride-2 transposed convolutions are
// decomposed by output parity so that no multiplication by a structural zero is ever issued.
//
// A workgroup owns a tile of BM = NB*TH*TW positions (NB whole images of TH x TW positions each) and BN
// output channels.  The input patch covering the tile (with halo) is staged ONCE per 32-channel chunk
// into LDS, and every tap reads its operand fragments from that patch at a shifted pixel offset.
#pragma once
#include "common.h"

#define CONV_MAX_TAPS 16
#define CONV_MAX_CLS 4
#define CONV_CK 32                 // input channels per staged chunk
#define CONV_RS 80                 // LDS row stride in bytes: 32 bf16 + 16 B pad (keeps 16-B alignment)

enum {
    CONV_3X3_S1 = 0,      // also the data-gradient of itself (with flipped, transposed weights)
    CONV_3X3_S2 = 1,
    CONV_1X1 = 2,         // also nn.Linear, and its own data-gradient (transposed weights)
    CONVT_2X2_S2 = 3,     // ConvTranspose2d(k=2, s=2): 4 parity classes of a 1x1
    DGRAD_3X3_S2 = 4,     // data-gradient of CONV_3X3_S2: 4 parity classes with 1/2/2/4 taps
    DGRAD_CONVT_2X2 = 5,  // data-gradient of CONVT_2X2_S2: a 2x2 stride-2 convolution
    CONV_KIND_COUNT = 6
};

enum { OUT_BF16_NHWC = 0, OUT_F32_NCHW = 1 };

struct ConvGeom {
    int N, IH, IW;         // input-side tensor (the one that is staged as patches)
    int OH, OW;            // output-side tensor
    int GH, GW;            // positions per image (per class)
    int SI, SO;            // input / output stride per position
    int dmin_h, dmin_w;    // input offset of a patch's first row / column relative to position*SI
    int dmax_h, dmax_w;    // largest tap offset relative to position*SI
    int PH, PW;            // patch rows / columns per image tile
    int TH, TW, NB;        // tile: NB images x TH x TW positions
    int tiles_h, tiles_w, tiles_n;
    int ncls;
    int cls_tap_begin[CONV_MAX_CLS + 1];
    int cls_off_h[CONV_MAX_CLS], cls_off_w[CONV_MAX_CLS];
    int tap_dh[CONV_MAX_TAPS], tap_dw[CONV_MAX_TAPS];   // patch-relative row / column offset of each tap (>= 0)
    int tap_poff[CONV_MAX_TAPS];    // dh*PW + dw : pixel offset inside the patch
    int tap_slice[CONV_MAX_TAPS];   // which [rows][Cin] weight slice the tap multiplies with
    int max_taps;                   // largest tap count of any class
    int nslices;
    int tw_shift, thw_shift;        // TW and TH*TW are powers of two
    unsigned pw_magic, phw_magic;   // x / PW = (x * pw_magic) >> 32 for x < 65536 (same for PH*PW): integer division costs ~40 instructions
};

static inline int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}
static inline unsigned div_magic(int d) { return d == 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }   // 0: divisor 1
#if defined(__HIPCC__)
__device__ __forceinline__ int div_fast(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }
#endif

static inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Fill the kind-dependent part of the geometry.  IH/IW are the dims of the staged (input-side) tensor.
// Returns 0 on success.
static inline int conv_geom_init(ConvGeom* g, int kind, int N, int IH, int IW, int bm) {
    g->N = N; g->IH = IH; g->IW = IW;
    g->ncls = 1;
    g->cls_off_h[0] = g->cls_off_w[0] = 0;
    int dh[CONV_MAX_TAPS], dw[CONV_MAX_TAPS];
    int nt = 0, dmax_h = 0, dmax_w = 0;
    g->dmin_h = g->dmin_w = 0;
    switch (kind) {
        case CONV_3X3_S1:
        case CONV_3X3_S2: {
            const int s = kind == CONV_3X3_S2 ? 2 : 1;
            g->SI = s; g->SO = 1;
            g->OH = (IH - 1) / s + 1; g->OW = (IW - 1) / s + 1;
            g->GH = g->OH; g->GW = g->OW;
            g->dmin_h = g->dmin_w = -1; dmax_h = dmax_w = 1;
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) { dh[nt] = kh; dw[nt] = kw; g->tap_slice[nt] = kh * 3 + kw; ++nt; }
            g->cls_tap_begin[0] = 0; g->cls_tap_begin[1] = 9;
            g->nslices = 9;
            break;
        }
        case CONV_1X1:
            g->SI = g->SO = 1; g->OH = IH; g->OW = IW; g->GH = IH; g->GW = IW;
            dh[0] = dw[0] = 0; g->tap_slice[0] = 0; nt = 1;
            g->cls_tap_begin[0] = 0; g->cls_tap_begin[1] = 1;
            g->nslices = 1;
            break;
        case CONVT_2X2_S2:
            g->SI = 1; g->SO = 2; g->OH = 2 * IH; g->OW = 2 * IW; g->GH = IH; g->GW = IW;
            g->ncls = 4;
            for (int c = 0; c < 4; ++c) {
                dh[c] = dw[c] = 0; g->tap_slice[c] = c;
                g->cls_tap_begin[c] = c; g->cls_off_h[c] = c >> 1; g->cls_off_w[c] = c & 1;
            }
            g->cls_tap_begin[4] = 4; nt = 4;
            g->nslices = 4;
            break;
        case DGRAD_3X3_S2: {
            // staged tensor = dY (IH x IW); output = dX (2*IH x 2*IW).  dX[2i+ph] gathers dY rows:
            //   ph = 0: kh = 1 -> row i ;  ph = 1: kh = 0 -> row i+1, kh = 2 -> row i.
            g->SI = 1; g->SO = 2; g->OH = 2 * IH; g->OW = 2 * IW; g->GH = IH; g->GW = IW;
            g->ncls = 4; dmax_h = dmax_w = 1;
            for (int c = 0; c < 4; ++c) {
                const int ph = c >> 1, pw = c & 1;
                g->cls_tap_begin[c] = nt; g->cls_off_h[c] = ph; g->cls_off_w[c] = pw;
                const int khs[2] = {ph ? 0 : 1, 2}, kws[2] = {pw ? 0 : 1, 2};
                for (int a = 0; a < (ph ? 2 : 1); ++a)
                    for (int b = 0; b < (pw ? 2 : 1); ++b) {
                        const int kh = khs[a], kw = kws[b];
                        dh[nt] = (ph && kh == 0) ? 1 : 0; dw[nt] = (pw && kw == 0) ? 1 : 0;
                        g->tap_slice[nt] = kh * 3 + kw; ++nt;
                    }
            }
            g->cls_tap_begin[4] = nt;
            g->nslices = 9;
            break;
        }
        case DGRAD_CONVT_2X2:
            g->SI = 2; g->SO = 1; g->OH = IH / 2; g->OW = IW / 2; g->GH = g->OH; g->GW = g->OW;
            dmax_h = dmax_w = 1;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) { dh[nt] = a; dw[nt] = b; g->tap_slice[nt] = a * 2 + b; ++nt; }
            g->cls_tap_begin[0] = 0; g->cls_tap_begin[1] = 4;
            g->nslices = 4;
            break;
        default: return -1;
    }
    g->max_taps = 0;
    for (int c = 0; c < g->ncls; ++c) {
        const int k = g->cls_tap_begin[c + 1] - g->cls_tap_begin[c];
        if (k > g->max_taps) g->max_taps = k;
    }
    // tile: TW columns (power of two, <= 16), TH rows, NB images, TH*TW*NB == bm
    int tw = next_pow2(g->GW); if (tw > 16) tw = 16; if (tw > bm) tw = bm;
    int th = next_pow2(g->GH); if (th > bm / tw) th = bm / tw;
    g->TW = tw; g->TH = th; g->NB = bm / (tw * th);
    g->PH = (th - 1) * g->SI + (dmax_h - g->dmin_h) + 1;
    g->PW = (tw - 1) * g->SI + (dmax_w - g->dmin_w) + 1;
    g->tiles_h = (g->GH + th - 1) / th; g->tiles_w = (g->GW + tw - 1) / tw; g->tiles_n = (N + g->NB - 1) / g->NB;
    for (int j = 0; j < nt; ++j) { g->tap_dh[j] = dh[j]; g->tap_dw[j] = dw[j]; g->tap_poff[j] = dh[j] * g->PW + dw[j]; }   // dh/dw are patch-relative (>= 0)
    g->dmax_h = dmax_h; g->dmax_w = dmax_w;
    g->tw_shift = ilog2_exact(tw); g->thw_shift = ilog2_exact(tw * th);
    g->pw_magic = div_magic(g->PW); g->phw_magic = div_magic(g->PH * g->PW);
    if (g->NB * g->PH * g->PW >= 65536) return -1;                          // range of the multiply-high division
    return 0;
}
