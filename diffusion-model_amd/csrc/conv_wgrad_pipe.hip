// Weight gradient of the deep, wide 3x3 stride-1 convolutions at large batch, as a pipeline on the matrix cores.
//
//   dW[ky][kx][co][ci] = sum over positions (n, y, x) of  dY[n][y][x][co] * X[n][y + ky - 1][x + kx - 1][ci]
//
// The tap-list kernel of conv_wgrad.hip stages one 128-position tile at a time (tables, loads, barrier, multiply, barrier)
// and owns 64 x 32 channels per workgroup, so every operand is re-read many times and the matrix cores wait for every tile.
// Here a 512-thread workgroup owns 128 output x 64 input channels x all nine taps (8 waves, each nine 32 x 32 fp32
// accumulator tiles) and a contiguous range of 64-position chunks (whole 8x8 images, four 4x4 images, or four rows of a
// 16x16 image).  A chunk's dY rows and its zero-haloed input patch sit in one of two LDS stages; while the waves multiply
// chunk c, the global loads of chunk c + 2 are in flight and chunk c + 1 is already in the other stage: one barrier per
// chunk.  Both operands are pixel-major in memory, i.e. [k][row] for the matrix instruction, and are fetched with the
// transposing LDS read.  All LDS addresses are a per-lane base plus compile-time offsets (the geometry is a template
// argument), so the inner loop is 80 reads and 36 matrix instructions with no address arithmetic.
// Output: per position split either its own fp32 copy of the weight block (`part`, folded by ddpm_reduce_rows) or atomics.
// Reference operation: the weight half of torch's conv2d backward (conv_block.py / constituent_blocks.py Conv2d layers).
#include "common.h"
#include "ddpm_hip.h"

int ddpm_conv_pipe_mode_now();            // conv_pipe.hip: 0 by problem size, 1 wherever the shape allows, -1 never
int ddpm_conv_wgrad_c32_splits(const ddpm_wgrad_desc* d);                     // conv_wgrad_c32.hip: the thin convolutions of the image ends
int ddpm_conv_wgrad_c32_try(const ddpm_wgrad_desc* d, hipStream_t stream);

namespace {

constexpr int WT = 512;                   // threads
constexpr int COB = 128, CIB = 64;        // channels of dY / of X per workgroup
constexpr int CH = 64;                    // positions per chunk
constexpr int PS = CIB * 2 + 16;          // patch pixel stride in LDS (bytes): 144
constexpr int DS = COB * 2 + 32;          // dY row stride in LDS (bytes): 288 (row q of a transposing read lands 8 banks on)

template <int W> struct Geo;
template <> struct Geo<16> { static constexpr int PW = 18, NPIX = 6 * 18, XSLOTS = 2; };     // four rows of one image + halo rows
template <> struct Geo<8> { static constexpr int PW = 10, NPIX = 10 * 10, XSLOTS = 1; };      // one image
template <> struct Geo<4> { static constexpr int PW = 6, NPIX = 4 * 36, XSLOTS = 1; };        // four images

template <int W> constexpr int stage_bytes() { return Geo<W>::NPIX * PS + CH * DS; }

struct WpArgs {
    const bf16* x; const bf16* x1; const bf16* dy;      // x1: second source of a concatenated input (channels C0 ..), or null
    const float* in_add; const unsigned char* in_mask; float in_mask_scale;      // the forward loader's transforms of source 0
    float* dw; float* part; long long dw_numel;
    int Cin, Cout, C0;                                  // Cin = C0 + C1
    int nchunks, splits, tiles;           // 64-position chunks in all; position splits (multiple of 8); channel tiles
    int ncib;                             // input-channel blocks (tile = cob * ncib + cib)
};

__device__ __forceinline__ bf16x8 tr_frag2(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

// patch pixel of position k of a chunk, and the pixel distance between the two rows of one transposing read (k, k + 4)
template <int W> __device__ __forceinline__ constexpr int pix_of(int k) {
    if constexpr (W == 16) return ((k >> 4) + 1) * 18 + (k & 15) + 1;
    else if constexpr (W == 8) return ((k >> 3) + 1) * 10 + (k & 7) + 1;
    else return (k >> 4) * 36 + (((k >> 2) & 3) + 1) * 6 + (k & 3) + 1;
}

template <int W, bool XF>
__global__ __launch_bounds__(WT) void conv_wgrad_pipe_kernel(const WpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = Geo<W>;
    constexpr int STAGE = stage_bytes<W>();
    constexpr int PATCH = G::NPIX * PS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroups of one position split sit on one XCD (ids that agree modulo 8): the tiles that share its rows share an L2
    const int id = blockIdx.x;
    const int split = (id / (8 * a.tiles)) * 8 + (id & 7);
    const int tile = (id % (8 * a.tiles)) >> 3;
    const int cob = tile / a.ncib, cib = tile - cob * a.ncib;
    const int co0 = cob * COB, ci0 = cib * CIB;
    const int c_begin = (int)((long long)a.nchunks * split / a.splits), c_end = (int)((long long)a.nchunks * (split + 1) / a.splits);

    // ---- zero both stages once: the halo pixels are never written again (except the 16x16 halo rows, written every chunk)
    for (int i = tid; i < 2 * STAGE / 16; i += WT) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(smem + i * 16) = z;
    }

    // ---- this thread's slots of a chunk (the same for every chunk): global element offset from the chunk's first row, LDS offset
    int d_goff[2], d_lds[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + u * WT, row = idx >> 4, vec = idx & 15;
        d_goff[u] = row * a.Cout + co0 + vec * 8;
        d_lds[u] = PATCH + row * DS + vec * 16;
    }
    // the input-channel block lies in one of the two sources of a concatenated input (both are multiples of the block)
    const bool second = ci0 >= a.C0;
    const bf16* const xsrc = second ? a.x1 : a.x;
    const int xcs = second ? a.Cin - a.C0 : a.C0, xc0 = second ? ci0 - a.C0 : ci0;
    int x_goff[G::XSLOTS], x_lds[G::XSLOTS], x_prow[G::XSLOTS];
#pragma unroll
    for (int u = 0; u < G::XSLOTS; ++u) {
        const int idx = tid + u * WT;
        if constexpr (W == 16) {
            const int prow = idx >> 7, px = (idx >> 3) & 15, vec = idx & 7;        // six patch rows (image rows y0 - 1 .. y0 + 4) of 16 pixels
            x_prow[u] = idx < 768 ? prow : -1;
            x_goff[u] = ((prow - 1) * 16 + px) * xcs + xc0 + vec * 8;
            x_lds[u] = (prow * 18 + px + 1) * PS + vec * 16;
        } else {
            const int pos = idx >> 3, vec = idx & 7;
            x_prow[u] = 1;
            x_goff[u] = pos * xcs + xc0 + vec * 8;
            x_lds[u] = pix_of<W>(pos) * PS + vec * 16;
        }
    }
    // the forward pass's loader transforms of source 0 (time-embedding add, then dropout mask: conv_mfma.hip), applied to the
    // fetched vector exactly as there: each rounds to bf16
    const bool xform = XF && !second;
    int x_img[G::XSLOTS];                               // image of the slot's pixel, relative to the chunk's first image
#pragma unroll
    for (int u = 0; u < G::XSLOTS; ++u) x_img[u] = W == 4 ? ((tid + u * WT) >> 3) >> 4 : 0;
    bf16x8 dv[2], xv[G::XSLOTS];
    f32x4 xa[G::XSLOTS][2];
    unsigned xm[G::XSLOTS];
    auto fetch = [&](int c) {
        const bf16* db = a.dy + (long long)c * CH * a.Cout;
        const bf16* xb = xsrc + (long long)c * CH * xcs;
        const long long img0 = W == 16 ? c >> 2 : W == 8 ? c : 4ll * c;
#pragma unroll
        for (int u = 0; u < 2; ++u) dv[u] = *(const bf16x8*)(db + d_goff[u]);
#pragma unroll
        for (int u = 0; u < G::XSLOTS; ++u) {
            bool ok = x_prow[u] >= 0;
            if constexpr (W == 16) {
                const int q = c & 3;                                  // which quarter of its image this chunk is
                ok = ok && (x_prow[u] != 0 || q != 0) && (x_prow[u] != 5 || q != 3);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[u][j] = (bf16)0.f;
            if (ok) xv[u] = *(const bf16x8*)(xb + x_goff[u]);
            if constexpr (XF) {
                xm[u] = 0xFFu;
#pragma unroll
                for (int j = 0; j < 4; ++j) { xa[u][0][j] = 0.f; xa[u][1][j] = 0.f; }
                if (xform && ok) {
                    if (a.in_add) {
                        const float* ap = a.in_add + (img0 + x_img[u]) * a.C0 + xc0 + ((tid + u * WT) & 7) * 8;
                        xa[u][0] = *(const f32x4*)ap; xa[u][1] = *(const f32x4*)(ap + 4);
                    }
                    if (a.in_mask) xm[u] = a.in_mask[((long long)c * CH * xcs + x_goff[u]) >> 3];
                }
            }
        }
    };
    auto stash = [&](char* st) {
#pragma unroll
        for (int u = 0; u < 2; ++u) *(bf16x8*)(st + d_lds[u]) = dv[u];
#pragma unroll
        for (int u = 0; u < G::XSLOTS; ++u) {
            if constexpr (XF) {
                if (xform) {          // (vectors of pixels outside the image were fetched as zeros with a zero add and a full mask: they stay zero)
                    if (a.in_add) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { xv[u][j] = (bf16)((float)xv[u][j] + xa[u][0][j]); xv[u][4 + j] = (bf16)((float)xv[u][4 + j] + xa[u][1][j]); }
                    }
                    if (a.in_mask) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[u][j] = (xm[u] >> j) & 1 ? (bf16)((float)xv[u][j] * a.in_mask_scale) : (bf16)0.f;
                    }
                }
            }
            if (x_prow[u] >= 0) *(bf16x8*)(st + x_lds[u]) = xv[u];
        }
    };

    // ---- lane-constant parts of the transposing reads
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int kq = 8 * (grp >> 1) + q;                 // position supplied by this lane in the first read; + 4 in the second
    const int colb = (16 * (grp & 1) + 4 * pp) * 2;    // byte offset of its 4 columns inside a 32-channel block
    const int wco = wave & 3, wci = wave >> 2;         // this wave's 32 output / 32 input channels of the workgroup's tile
    const int a_lane = PATCH + kq * DS + wco * 64 + colb;
    const int b_lane = pix_of<W>(kq) * PS + wci * 64 + colb;
    constexpr int P1 = (W == 4 ? 6 : 4) * PS;          // position k + 4: four pixels on in the row (or, 4x4 images, the next row)

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    __syncthreads();
    if (c_begin < c_end) { fetch(c_begin); stash(smem); }
    if (c_begin + 1 < c_end) fetch(c_begin + 1);
    __syncthreads();

    for (int c = c_begin; c < c_end; ++c) {
        const int par = (c - c_begin) & 1;
        char* cur = smem + par * STAGE;
        if (c + 1 < c_end) stash(smem + (par ^ 1) * STAGE);           // chunk c + 1 (loaded during the previous chunk's multiplications)
        if (c + 2 < c_end) fetch(c + 2);                              // in flight during this chunk's multiplications
        const char* pa = cur + a_lane;
        const char* pb = cur + b_lane;
#pragma unroll
        for (int ks = 0; ks < CH / 16; ++ks) {
            const bf16x8 av = tr_frag2(pa + ks * 16 * DS, pa + ks * 16 * DS + 4 * DS);
            const int koff = (pix_of<W>(ks * 16) - pix_of<W>(0)) * PS;   // compile-time after unrolling
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = ((t / 3 - 1) * G::PW + (t % 3 - 1)) * PS;
                const bf16x8 bv = tr_frag2(pb + koff + toff, pb + koff + toff + P1);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- flush: rows = output channels, columns = input channels
    const int ci = ci0 + wci * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const long long soff = (long long)t * a.Cout * a.Cin;
        if (a.part) {
            float* base = a.part + (long long)split * a.dw_numel + soff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                base[(long long)co * a.Cin + ci] = acc[t][r];
            }
        } else {
            float* base = a.dw + soff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                atomicAdd(base + (long long)co * a.Cin + ci, acc[t][r]);
            }
        }
    }
}

template <int W, bool XF>
int launch_as(const WpArgs& a, hipStream_t stream) {
    constexpr int lds = 2 * stage_bytes<W>();
    static_assert(lds <= 160 * 1024, "stages do not fit LDS");
    auto k = conv_wgrad_pipe_kernel<W, XF>;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            ddpm_set_error("wgrad (pipelined): cannot reserve %d bytes of LDS", lds);
            return DDPM_EHIP;
        }
        attr = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(a.splits * a.tiles)), dim3(WT), lds, stream, a);
    return ddpm_check_launch("conv_wgrad_pipe");
}

template <int W>
int launch(const WpArgs& a, hipStream_t stream) {
    return (a.in_add || a.in_mask) ? launch_as<W, true>(a, stream) : launch_as<W, false>(a, stream);
}

// shape part of the decision (the descriptor's pointers are not looked at): fills the split / tile counts
bool plan(const ddpm_wgrad_desc* d, WpArgs& a) {
    const int mode = ddpm_conv_pipe_mode_now();
    if (mode < 0) return false;
    if (d->kind != DDPM_CONV_3X3_S1 || d->tiles_per_split > 0) return false;    // (grouped launches size their partial rows for the tap-list kernel)
    if (d->IH != d->IW || (d->IW != 4 && d->IW != 8 && d->IW != 16)) return false;
    if (d->C0 % CIB || d->C1 % CIB || d->Cout % COB || d->dy_cs != d->Cout) return false;
    const long long rows = (long long)d->N * d->IH * d->IW;
    if (rows % CH || rows / CH > 0x7fffffffll / CH) return false;
    a.Cin = d->C0 + d->C1; a.Cout = d->Cout; a.C0 = d->C0;
    a.nchunks = (int)(rows / CH);
    a.ncib = a.Cin / CIB;
    a.tiles = (d->Cout / COB) * a.ncib;
    int splits = (256 / a.tiles) & ~7;                  // one workgroup per CU; a multiple of 8 (one XCD per split)
    if (splits < 8) splits = 8;
    // by size: every split should run long enough (>= 16 chunks) for the fill, the flush and the fold of its partial block to vanish
    if (mode == 0 && a.nchunks < 16 * splits) return false;
    while (splits > 8 && a.nchunks < splits) splits -= 8;
    if (a.nchunks < splits) return false;
    a.splits = splits;
    a.dw_numel = 9ll * d->Cout * a.Cin;
    return true;
}

}  // namespace

// Number of `part` rows the pipelined kernel writes for this problem, or 0 when it does not take it.
extern "C" int ddpm_conv_wgrad_pipe_splits(const ddpm_wgrad_desc* d) {
    if (const int s = ddpm_conv_wgrad_c32_splits(d)) return s;
    WpArgs a;
    return plan(d, a) ? a.splits : 0;
}

// Returns 1 when the problem was launched here, 0 when the caller should use another kernel, < 0 on error.
int ddpm_conv_wgrad_pipe_try(const ddpm_wgrad_desc* d, hipStream_t stream) {
    if (const int rc = ddpm_conv_wgrad_c32_try(d, stream)) return rc;
    WpArgs a;
    if (!plan(d, a)) return 0;
    a.x = (const bf16*)d->in0; a.x1 = (const bf16*)d->in1; a.dy = (const bf16*)d->dy; a.dw = d->dw; a.part = d->part;
    a.in_add = d->in_add; a.in_mask = d->in_mask; a.in_mask_scale = d->in_mask_scale;
    const int rc = d->IW == 16 ? launch<16>(a, stream) : d->IW == 8 ? launch<8>(a, stream) : launch<4>(a, stream);
    return rc == DDPM_OK ? 1 : -1;
}
