// Weight gradient of the dense convolutions on bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   dW[slice_j][co][ci] += sum over positions p of  dY[out pixel of p][co] * X[in pixel of p shifted by tap j][ci]
//
// GEMM view: the contraction runs over POSITIONS (N*GH*GW, the long dimension), the output is the small
// [taps][Cout][Cin] weight block.  Both operands are stored pixel-major (NHWC), i.e. [k][row] from the
// MFMA's point of view, so their fragments are fetched from LDS with the transposing read
// ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group, delivered column-major).
//
// A workgroup owns (32*NCO output channels) x (32 input channels) x (all taps of its class) and walks a
// strided subset of the position tiles.  Per tile it stages the dY tile and the input patch (with halo,
// with the same fused input transforms as the forward kernel: concat, time-embedding add, dropout mask)
// once; the (tap, co-subtile) work items are dealt round-robin to the 4 waves, each keeping its items'
// 32x32 fp32 accumulators in registers across all tiles.  One atomic flush per workgroup at the end.
#include "conv_geom.h"
#include "ddpm_hip.h"

struct WgradArgs {
    ConvGeom g;
    const bf16* in0; const bf16* in1; int C0, C1;
    const float* in_add; const unsigned char* in_mask; float in_mask_scale;
    const bf16* dy; int Cout, dy_cs;
    float* dw;
    float* part;              // when set: per-split partial sums [splits][slices][Cout][Cin], plain stores, no atomics
    long long dw_numel;
    int ntiles;
    int gx, gy;               // grid of this problem: gx position splits x gy (output, input) channel blocks x ncls classes
};

__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
    // two transposed 4x16 block reads -> the 8 k-values of this lane's row/column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(bf16x8, r);
}

#define WG_U 4

template <int NCO, int MAXI>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bx, const int by, const int bz, const int gx, char* smem) {
    constexpr int T = 256, BM = 128;
    constexpr int DYS = NCO * 64 + 16;          // dY tile row stride (bytes)
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cls = bz;
    const int tap0 = g.cls_tap_begin[cls], ntaps = g.cls_tap_begin[cls + 1] - tap0;
    const int Cin = a.C0 + a.C1;
    const int nci = (Cin + CONV_CK - 1) / CONV_CK;
    const int cob = by / nci, cib = by - cob * nci;
    const int co0 = cob * 32 * NCO, c0 = cib * CONV_CK;
    const int parts = (Cin - c0 >= CONV_CK) ? 4 : (Cin - c0) / 8;
    const int PHW = g.PH * g.PW, npix = g.NB * PHW;
    const int thw = g.TH * g.TW;
    const int offh = g.cls_off_h[cls], offw = g.cls_off_w[cls];

    int* s_pl = (int*)smem;                                        // patch slot -> linear input pixel or -1
    int* s_pb = s_pl + ((npix + 3) & ~3);                          // position -> patch pixel base
    int* s_po = s_pb + BM;                                         // position -> linear output pixel or -1
    char* s_patch = (char*)(s_po + BM);
    char* s_dy = s_patch + npix * CONV_RS;

    for (int p = tid; p < BM; p += T) {
        const int il = p / thw, rem = p - il * thw;
        const int ty = rem / g.TW, tx = rem - ty * g.TW;
        s_pb[p] = il * PHW + ty * g.SI * g.PW + tx * g.SI;
    }

    const bool src1 = c0 >= a.C0;
    const bf16* src = src1 ? a.in1 : a.in0;
    const int cs = src1 ? a.C1 : a.C0, cc = src1 ? c0 - a.C0 : c0;
    const int ihw = g.IH * g.IW;

    // work items of this wave: item = tap * NCO + co-subtile, dealt round-robin
    const int nitems = ntaps * NCO;
    f32x16 acc[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // lane-constant parts of the transposed-read addresses
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int kq = 8 * (grp >> 1) + q;                 // k row supplied by this lane (first read); +4 for the second
    const int colb = (16 * (grp & 1) + 4 * pp) * 2;    // byte offset of its 4 columns inside a 32-channel row

    for (int tile = bx; tile < a.ntiles; tile += gx) {
        int t = tile;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h;
        const int tni = t / g.tiles_h;
        const int n0 = tni * g.NB, oy = thi * g.TH, ox = twi * g.TW;
        __syncthreads();     // previous tile fully consumed
        for (int p = tid; p < npix; p += T) {
            const int il = p / PHW, r = p - il * PHW;
            const int py = r / g.PW, px = r - py * g.PW;
            const int n = n0 + il, iy = oy * g.SI + g.dmin_h + py, ix = ox * g.SI + g.dmin_w + px;
            const bool ok = n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
            s_pl[p] = ok ? (n * g.IH + iy) * g.IW + ix : -1;
        }
        for (int p = tid; p < BM; p += T) {
            const int il = p / thw, rem = p - il * thw;
            const int ty = rem / g.TW, tx = rem - ty * g.TW;
            const int n = n0 + il, py = oy + ty, px = ox + tx;
            const int oh = py * g.SO + offh, ow = px * g.SO + offw;
            const bool ok = n < g.N && py < g.GH && px < g.GW && oh < g.OH && ow < g.OW;
            s_po[p] = ok ? (n * g.OH + oh) * g.OW + ow : -1;
        }
        __syncthreads();
        // -- stage the input patch (32 channels of this block's chunk) and the dY tile (BM positions x 32*NCO output
        //    channels, zero rows for masked positions) in one batched round: all global loads first, then LDS writes.
        const int n_patch = npix * 4, n_items = n_patch + BM * NCO * 4;
        const bool xform = !src1 && (a.in_add || a.in_mask);
        for (int base = tid; base < n_items; base += T * WG_U) {
            bf16x8 v[WG_U];
            long long e[WG_U];
            int pl[WG_U], ldst[WG_U];
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                const int it = base + u * T;
                pl[u] = -1; ldst[u] = -1; e[u] = 0;
                if (it < n_patch) {
                    const int p = it >> 2, part = it & 3;
                    if (part < parts) {
                        pl[u] = s_pl[p];
                        ldst[u] = p * CONV_RS + part * 16;
                        e[u] = (long long)pl[u] * cs + cc + part * 8;
                    }
                } else if (it < n_items) {
                    const int di = it - n_patch;
                    const int p = di / (NCO * 4), part = di - p * (NCO * 4);
                    const int po = s_po[p];
                    const int co = co0 + part * 8;
                    ldst[u] = (int)(s_dy - s_patch) + p * DYS + part * 16;
                    if (po >= 0 && co + 8 <= a.dy_cs) { pl[u] = -2; e[u] = (long long)po * a.dy_cs + co; }
                }
            }
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
                if (pl[u] >= 0) v[u] = *(const bf16x8*)(src + e[u]);
                else if (pl[u] == -2) v[u] = *(const bf16x8*)(a.dy + e[u]);
            }
            if (xform) {
#pragma unroll
                for (int u = 0; u < WG_U; ++u) {
                    if (pl[u] < 0) continue;
                    const int part = (base + u * T) & 3;
                    if (a.in_add) {
                        const float* ad = a.in_add + (pl[u] / ihw) * a.C0 + cc + part * 8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[u][j] = (bf16)((float)v[u][j] + ad[j]);
                    }
                    if (a.in_mask) {
                        const unsigned m = a.in_mask[e[u] >> 3];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[u][j] = (m >> j) & 1 ? (bf16)((float)v[u][j] * a.in_mask_scale) : (bf16)0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < WG_U; ++u)
                if (ldst[u] >= 0) *(bf16x8*)(s_patch + ldst[u]) = v[u];
        }
        __syncthreads();
        // -- MFMA: for each of this wave's items, contract over the tile's 128 positions (8 k-steps of 16)
        if (parts >= 2) {
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int item = wave + 4 * i;
                if (item < nitems) {
                    const int j = item / NCO, cosub = item - j * NCO;
                    const int poff = g.tap_poff[tap0 + j] * CONV_RS;
                    const char* dyp = s_dy + cosub * 64 + colb;
                    for (int ks = 0; ks < BM / 16; ++ks) {
                        const int k = ks * 16 + kq;
                        const bf16x8 av = tr_frag(dyp + k * DYS, dyp + (k + 4) * DYS);
                        const bf16x8 bv = tr_frag(s_patch + s_pb[k] * CONV_RS + poff + colb, s_patch + s_pb[k + 4] * CONV_RS + poff + colb);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- flush: dw[slice][co][ci] += acc  (rows = co, columns = ci)
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int item = wave + 4 * i;
        if (item >= nitems) continue;
        const int j = item / NCO, cosub = item - j * NCO;
        const int ci = c0 + (lane & 31);
        const long long soff = (long long)g.tap_slice[tap0 + j] * a.Cout * Cin;
        if (a.part) {
            // two-stage reduction: the position splits of one weight tile would otherwise contend with atomics on the
            // same few cache lines; each split stores its tile, one table-driven pass folds all layers later
            float* base = a.part + (long long)bx * a.dw_numel + soff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cosub * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < a.Cout && ci < Cin) base[(long long)co * Cin + ci] = acc[i][r];
            }
        } else {
            float* base = a.dw + soff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cosub * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < a.Cout && ci < Cin) atomicAdd(base + (long long)co * Cin + ci, acc[i][r]);
            }
        }
    }
}

template <int NCO, int MAXI>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_body<NCO, MAXI>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, smem);
}

// Many independent weight-gradient problems in ONE launch: a weight gradient is a leaf of the backward pass, so all
// of a network's can run together at its end, filling the machine instead of paying a latency-bound launch each.
// tab: the problems of this kernel variant; first[p] = first flat workgroup of problem p (first[n] = total).
template <int NCO, int MAXI>
__global__ __launch_bounds__(256, 2) void conv_wgrad_group_kernel(const WgradArgs* __restrict__ tab, const int* __restrict__ first, const int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first[mid] <= b) lo = mid; else hi = mid;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    // the problem description moves to LDS once: read through a reference into global memory it would be fetched
    // again after every store (the tap tables are indexed at run time, so registers are no option either)
    constexpr int ARGS_BYTES = (sizeof(WgradArgs) + 15) & ~15;
    {
        const int* src = (const int*)(tab + lo);
        int* dst = (int*)smem;
        for (int i = threadIdx.x; i < (int)(sizeof(WgradArgs) / 4); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const WgradArgs& a = *(const WgradArgs*)smem;
    const int local = b - first[lo];
    const int gx = a.gx, gy = a.gy;
    const int bx = local % gx, r = local / gx;
    wgrad_body<NCO, MAXI>(a, bx, r % gy, r / gy, gx, smem + ARGS_BYTES);
}


// ---- pointwise (1x1 / Linear) weight gradient ---------------------------------------------------------------------
// dW[co][ci] = sum over rows of dY[row][co] * X[row][ci]: no patches, no halo, rows contiguous in memory.  Two thirds
// of the network's weight-gradient problems have this form, and for them the tap-list kernel's per-tile pixel tables
// and single-buffered staging are pure overhead.  A workgroup owns up to 128 output x 128 input channels (16 tiles of
// 32x32 dealt to the 4 waves) and a contiguous range of rows, streamed in chunks of 64 through a double-buffered LDS
// stage: the global loads of chunk c+1 are in flight while the MFMAs of chunk c run, one barrier per chunk.
struct PwArgs {
    const bf16* x; const bf16* dy; const unsigned char* in_mask; float in_mask_scale;
    float* dw; float* part; long long dw_numel;
    long long R, rows_per_split;
    int Cin, Cout, dy_cs;
    int gx, nib;              // position splits; input-channel blocks (blockIdx.y = cb * nib + ib)
};

#define PW_ROWS 64
#define PW_RS 272             // LDS row stride in bytes: 128 channels + 16 bytes of padding
#define PW_RS_NARROW 144      // the same for problems of at most 64 x 64 channels (64 channels + 16 bytes of padding)
#define PW_NARROW_MAX 64

// NACC: 32x32 tile pairs per wave (4 covers 128 x 128 channels, 1 covers 64 x 64); NSLOT: 16-byte vectors per thread and chunk
// (8 covers 64 rows of 128 + 128 channels, 4 covers 64 + 64).  The narrow form needs a third of the registers.
template <int RS, int NACC, int NSLOT>
__device__ __forceinline__ void pw_body(const PwArgs& a, const int bx, const int by, char* smem) {
    constexpr int CB = RS == PW_RS_NARROW ? PW_NARROW_MAX : 128;      // channels of either operand per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = by / a.nib, ib = by - cb * a.nib;
    const int co0 = cb * CB, ci0 = ib * CB;
    const int cow = min(CB, ((a.Cout - co0) + 31) & ~31), ciw = min(CB, a.Cin - ci0);
    const int nct = cow / 32, nit = ciw / 32, npairs = nct * nit;
    const int xv = ciw / 8, dv = cow / 8;                   // 16-byte vectors per row of each operand
    const int nx = PW_ROWS * xv, nitems = nx + PW_ROWS * dv; // <= 2048 vectors per chunk
    const long long r_begin = (long long)bx * a.rows_per_split;
    const long long r_end = min(a.R, r_begin + a.rows_per_split);

    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int kq = 8 * (grp >> 1) + q;
    const int colb = (16 * (grp & 1) + 4 * pp) * 2;

    // per-thread slots of a chunk (the same for every chunk): row, global element offset inside the chunk, LDS offset;
    // computed once - the integer divisions would otherwise cost more than the MFMAs of a chunk
    bf16x8 v[NSLOT];
    int s_row[NSLOT], s_goff[NSLOT], s_lds[NSLOT];
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
        const int it = tid + u * 256;
        s_row[u] = -1; s_goff[u] = 0; s_lds[u] = 0;
        if (it < nx) {
            const int row = it / xv, c = (it - row * xv) * 8;
            s_row[u] = row; s_goff[u] = row * a.Cin + ci0 + c; s_lds[u] = row * RS + c * 2;
        } else if (it < nitems) {
            const int di = it - nx;
            const int row = di / dv, c = (di - row * dv) * 8;
            if (co0 + c + 8 <= a.dy_cs) { s_row[u] = row | 0x10000; s_goff[u] = row * a.dy_cs + co0 + c; }
            else s_row[u] = -2;                                    // padded output channels: zeros
            s_lds[u] = (PW_ROWS + row) * RS + c * 2;
        }
    }
    const bf16* xg = a.x;
    const bf16* dg = a.dy;
    const unsigned char* mg = a.in_mask;
    const float mscale = a.in_mask_scale;
    const long long x_rs = a.Cin, d_rs = a.dy_cs;
    auto fetch = [&](long long r0) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
            if (s_row[u] < 0) continue;
            const int row = s_row[u] & 0xFFFF;
            if (r0 + row >= r_end) continue;
            if (s_row[u] & 0x10000) {
                v[u] = *(const bf16x8*)(dg + r0 * d_rs + s_goff[u]);
            } else {
                const long long e = r0 * x_rs + s_goff[u];
                v[u] = *(const bf16x8*)(xg + e);
                if (mg) {
                    const unsigned m = mg[e >> 3];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] = (m >> j) & 1 ? (bf16)((float)v[u][j] * mscale) : (bf16)0.f;
                }
            }
        }
    };
    auto stash = [&](char* b) {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u)
            if (s_row[u] != -1) *(bf16x8*)(b + s_lds[u]) = v[u];
    };

    if (r_begin < r_end) fetch(r_begin);
    int par = 0;
    for (long long r0 = r_begin; r0 < r_end; r0 += PW_ROWS, par ^= 1) {
        char* b = smem + par * (2 * PW_ROWS * RS);       // each buffer: X rows, then dY rows
        stash(b);
        if (r0 + PW_ROWS < r_end) fetch(r0 + PW_ROWS);
        __syncthreads();
        const char* sx = b;
        const char* sd = b + PW_ROWS * RS;
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int pr = wave + 4 * i;
            if (pr < npairs) {
                const int ct = pr / nit, itile = pr - ct * nit;
                const char* dyp = sd + ct * 64 + colb;
                const char* xp = sx + itile * 64 + colb;
#pragma unroll
                for (int ks = 0; ks < PW_ROWS / 16; ++ks) {
                    const int k = ks * 16 + kq;
                    const bf16x8 av = tr_frag(dyp + k * RS, dyp + (k + 4) * RS);
                    const bf16x8 bv = tr_frag(xp + k * RS, xp + (k + 4) * RS);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int pr = wave + 4 * i;
        if (pr >= npairs) continue;
        const int ct = pr / nit, itile = pr - ct * nit;
        const int ci = ci0 + itile * 32 + (lane & 31);
        float* base = a.part ? a.part + (long long)bx * a.dw_numel : a.dw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (co < a.Cout && ci < a.Cin) {
                if (a.part) base[(long long)co * a.Cin + ci] = acc[i][r];
                else atomicAdd(base + (long long)co * a.Cin + ci, acc[i][r]);
            }
        }
    }
}

template <int RS>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(const PwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (RS == PW_RS_NARROW) pw_body<RS, 1, 4>(a, blockIdx.x, blockIdx.y, smem);
    else pw_body<RS, 4, 8>(a, blockIdx.x, blockIdx.y, smem);
}

// RS = PW_RS_NARROW: workgroups of 64 x 64 channels (a quarter of the registers, half the LDS: four workgroups per CU).  Two
// kinds of problem go there (wgrad_plan): the pointwise convolutions of the large-image stages (few channels, hundreds of
// thousands of rows), which are bound by the loads in flight; and wider problems whose 128 x 128 blocking would give fewer
// than 512 workgroups - a workgroup has one chunk of loads in flight, so a launch of 34 .. 144 workgroups (what the deep
// stages of a 32x32 network give) runs at the latency of its chunks while most CUs idle.  Four times the workgroups read each
// operand twice (from L2: these tensors are a few MB) and finish in about half the time.
template <int RS>
__global__ __launch_bounds__(256) void pw_wgrad_group_kernel(const PwArgs* __restrict__ tab, const int* __restrict__ first, const int n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (first[mid] <= b) lo = mid; else hi = mid;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const PwArgs a = tab[lo];          // by value: a reference into global memory would be re-read around every store
    const int local = b - first[lo];
    if constexpr (RS == PW_RS_NARROW) pw_body<RS, 1, 4>(a, local % a.gx, local / a.gx, smem);
    else pw_body<RS, 4, 8>(a, local % a.gx, local / a.gx, smem);
}

#define PW_LDS (4 * PW_ROWS * PW_RS)
#define PW_LDS_NARROW (4 * PW_ROWS * PW_RS_NARROW)

template <int NCO, int MAXI>
static int launch_wgrad(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    auto k = conv_wgrad_kernel<NCO, MAXI>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ddpm_check_launch("wgrad: raise dynamic LDS limit");
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, a);
    return ddpm_check_launch("conv_wgrad");
}

struct WgradPlan {
    WgradArgs a;
    PwArgs pw;
    bool pointwise, pw_narrow;
    int nco, maxi;
    dim3 grid;
    size_t lds;
};

static int wgrad_plan(const ddpm_wgrad_desc* d, WgradPlan& p) {
    DDPM_REQUIRE(d && d->in0 && d->dy && (d->dw || d->part), "wgrad: null pointer");
    DDPM_REQUIRE(d->kind == CONV_3X3_S1 || d->kind == CONV_3X3_S2 || d->kind == CONV_1X1 || d->kind == CONVT_2X2_S2,
                 "wgrad: kind %d is not a forward kind", d->kind);
    DDPM_REQUIRE(d->N > 0 && d->IH > 0 && d->IW > 0, "wgrad: bad dims");
    DDPM_REQUIRE(d->C0 > 0 && d->C0 % 16 == 0 && d->C1 >= 0 && d->C1 % 32 == 0 && (d->C1 == 0 || (d->in1 && d->C0 % 32 == 0)),
                 "wgrad: input channels must be multiples of 16 (32 when two sources are concatenated); got %d + %d", d->C0, d->C1);
    DDPM_REQUIRE(d->Cout > 0 && d->dy_cs >= d->Cout && d->dy_cs % 8 == 0, "wgrad: dy channel stride %d must be a multiple of 8 >= Cout %d",
                 d->dy_cs, d->Cout);
    WgradArgs& a = p.a;
    if (conv_geom_init(&a.g, d->kind, d->N, d->IH, d->IW, 128) != 0) { ddpm_set_error("wgrad: geometry"); return DDPM_EINVAL; }
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.C0 = d->C0; a.C1 = d->C1;
    a.in_add = d->in_add; a.in_mask = d->in_mask; a.in_mask_scale = d->in_mask_scale;
    a.dy = (const bf16*)d->dy; a.Cout = d->Cout; a.dy_cs = d->dy_cs; a.dw = d->dw; a.part = d->part;
    a.ntiles = a.g.tiles_n * a.g.tiles_h * a.g.tiles_w;
    const int Cin = d->C0 + d->C1;
    const int slices = d->kind == CONV_1X1 ? 1 : d->kind == CONVT_2X2_S2 ? 4 : 9;
    a.dw_numel = (long long)slices * d->Cout * Cin;
    p.pointwise = d->kind == CONV_1X1 && d->C1 == 0 && !d->in_add && Cin % 32 == 0;
    p.pw_narrow = false;
    if (p.pointwise) {
        PwArgs& w = p.pw;
        w.x = a.in0; w.dy = a.dy; w.in_mask = a.in_mask; w.in_mask_scale = a.in_mask_scale;
        w.dw = a.dw; w.part = a.part; w.dw_numel = a.dw_numel;
        w.R = (long long)d->N * d->IH * d->IW;
        w.Cin = Cin; w.Cout = d->Cout; w.dy_cs = d->dy_cs;
        const long long tiles = (w.R + 127) / 128;
        long long msplit = 0, other = 0;
        p.pw_narrow = Cin <= PW_NARROW_MAX && d->Cout <= PW_NARROW_MAX;
        for (int pass = 0; pass < 2; ++pass) {
            const int cb = p.pw_narrow ? PW_NARROW_MAX : 128;
            const int ncb = (d->Cout + cb - 1) / cb;
            w.nib = (Cin + cb - 1) / cb;
            other = (long long)ncb * w.nib;
            msplit = 512 / other;
            if (d->tiles_per_split > 0) {
                const long long want = (tiles + d->tiles_per_split - 1) / d->tiles_per_split;
                if (want < msplit) msplit = want;
            }
            if (msplit < 1) msplit = 1;
            if (msplit > tiles) msplit = tiles;
            w.rows_per_split = (tiles + msplit - 1) / msplit * 128;
            msplit = (w.R + w.rows_per_split - 1) / w.rows_per_split;
            // too few workgroups for the machine in blocks of 128 x 128 channels: blocks of 64 x 64 (see pw_wgrad_group_kernel)
            if (p.pw_narrow || msplit * other >= 512) break;
            p.pw_narrow = true;
        }
        w.gx = (int)msplit;
        p.grid = dim3((unsigned)msplit, (unsigned)other, 1);
        p.lds = p.pw_narrow ? PW_LDS_NARROW : PW_LDS;
        p.nco = p.maxi = 0;
        a.gx = (int)msplit; a.gy = (int)other;
        return DDPM_OK;
    }
    const int co_sub = (d->Cout + 31) / 32;
    int nco = a.g.max_taps >= 9 ? 2 : 4;
    if (nco > co_sub) nco = co_sub;
    if (nco == 3) nco = 2;
    const int cob = (co_sub + nco - 1) / nco, nci = (Cin + CONV_CK - 1) / CONV_CK;
    p.nco = nco;
    p.maxi = (a.g.max_taps * nco + 3) / 4;
    long long other = (long long)cob * nci * a.g.ncls;
    long long msplit = 512 / other;
    if (d->tiles_per_split > 0) {
        // grouped launches fill the machine with many problems at once: fewer, longer position splits mean fewer
        // partial tiles to store and to fold afterwards
        // (a tile of a stride-2 problem costs a workgroup about twice the time of any other tile - its patch holds four input
        // pixels per position - and these problems are few, small workgroup chains that decide how long their launch lasts:
        // half the tiles per split.  Measured on the whole step: 2.85 -> 2.82 ms; a quarter, or fewer tiles for the other
        // 9-tap problems as well, is slower again because of the partial tiles to store and fold)
        const int tps = d->kind == CONV_3X3_S2 ? (d->tiles_per_split + 1) / 2 : d->tiles_per_split;
        const long long want = (a.ntiles + tps - 1) / tps;
        if (want < msplit) msplit = want;
    }
    if (msplit < 1) msplit = 1;
    if (msplit > a.ntiles) msplit = a.ntiles;
    const int npix = a.g.NB * a.g.PH * a.g.PW;
    p.lds = ((size_t)((npix + 3) & ~3) + 128 + 128) * 4 + (size_t)npix * CONV_RS + (size_t)128 * (nco * 64 + 16);
    DDPM_REQUIRE(p.lds <= 160 * 1024, "wgrad: tile needs %zu B of LDS", p.lds);
    p.grid = dim3((unsigned)msplit, (unsigned)(cob * nci), (unsigned)a.g.ncls);
    a.gx = (int)msplit; a.gy = cob * nci;
    return DDPM_OK;
}

int ddpm_conv_wgrad_pipe_try(const ddpm_wgrad_desc* d, hipStream_t stream);

extern "C" int ddpm_conv_wgrad_splits(const ddpm_wgrad_desc* d) {
    if (const int s = ddpm_conv_wgrad_pipe_splits(d)) return s;
    WgradPlan p;
    ddpm_wgrad_desc tmp = *d;
    if (!tmp.dw && !tmp.part) tmp.part = (float*)16;        // sizing query: output pointers are not needed yet
    if (wgrad_plan(&tmp, p) != DDPM_OK) return -1;
    return (int)p.grid.x;
}

extern "C" int ddpm_conv_wgrad(const ddpm_wgrad_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WgradPlan p;
    if (int rc = wgrad_plan(d, p)) return rc;            // (also validates the descriptor)
    if (const int rc = ddpm_conv_wgrad_pipe_try(d, stream)) return rc > 0 ? DDPM_OK : DDPM_EHIP;
    if (p.pointwise) {
        if (p.pw_narrow) {
            hipLaunchKernelGGL(pw_wgrad_kernel<PW_RS_NARROW>, p.grid, dim3(256), PW_LDS_NARROW, stream, p.pw);
            return ddpm_check_launch("pw_wgrad (narrow)");
        }
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute((const void*)pw_wgrad_kernel<PW_RS>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS) != hipSuccess)
                return ddpm_check_launch("wgrad: raise dynamic LDS limit");
            attr = true;
        }
        hipLaunchKernelGGL(pw_wgrad_kernel<PW_RS>, p.grid, dim3(256), PW_LDS, stream, p.pw);
        return ddpm_check_launch("pw_wgrad");
    }
    const WgradArgs& a = p.a;
    const int nco = p.nco, maxi = p.maxi;
    const dim3 grid = p.grid;
    const size_t lds = p.lds;
#define WG_CASE(NCOv, MAXIv) if (nco == NCOv && maxi == MAXIv) return launch_wgrad<NCOv, MAXIv>(a, grid, lds, stream);
    WG_CASE(1, 1) WG_CASE(2, 1) WG_CASE(4, 1)      /* 1-tap classes: 1x1, convT */
    WG_CASE(1, 3) WG_CASE(2, 5)                    /* 9 taps */
#undef WG_CASE
    ddpm_set_error("wgrad: no kernel for nco=%d items/wave=%d", nco, maxi);
    return DDPM_EINVAL;
}

// ---- grouped launch ---------------------------------------------------------------------------------------------
// Blob layout (identical bytes on host and device): header, then per kernel variant the WgradArgs of its problems and
// the prefix table of their workgroup counts.
#define WG_VARIANTS 7          // five tap-list kernel instances + the pointwise kernel (index 5) + its narrow form (index 6)
static const int kVariantNco[WG_VARIANTS] = {1, 2, 4, 1, 2, 0, -1};
static const int kVariantMaxi[WG_VARIANTS] = {1, 1, 1, 3, 5, 0, -1};

struct WgradGroupHeader {
    long long magic;
    struct V { long long args_off, first_off, lds; int nprob, total_blocks; } v[WG_VARIANTS];
};

static inline long long align16(long long v) { return (v + 15) & ~15LL; }

extern "C" int64_t ddpm_conv_wgrad_group_bytes(int n) {
    if (n < 0) n = 0;
    return align16(sizeof(WgradGroupHeader)) + WG_VARIANTS * 32 + align16((long long)n * sizeof(WgradArgs)) +
           align16((long long)(n + WG_VARIANTS) * sizeof(int)) + 64;
}

extern "C" int ddpm_conv_wgrad_group_plan(const ddpm_wgrad_desc* descs, int n, void* host_blob, int64_t bytes) {
    DDPM_REQUIRE(descs && host_blob && n > 0, "wgrad group: bad arguments");
    DDPM_REQUIRE(bytes >= ddpm_conv_wgrad_group_bytes(n), "wgrad group: blob of %lld bytes is too small for %d problems", (long long)bytes, n);
    char* base = (char*)host_blob;
    WgradGroupHeader* h = (WgradGroupHeader*)base;
    h->magic = 0x5747524144LL;
    static thread_local WgradPlan plans[1024];
    DDPM_REQUIRE(n <= 1024, "wgrad group: at most 1024 problems per group");
    int variant_of[1024];
    for (int i = 0; i < n; ++i) {
        if (int rc = wgrad_plan(&descs[i], plans[i])) return rc;
        variant_of[i] = -1;
        for (int v = 0; v < WG_VARIANTS; ++v)
            if (plans[i].nco == kVariantNco[v] && plans[i].maxi == kVariantMaxi[v]) variant_of[i] = v;
        if (plans[i].pointwise) variant_of[i] = plans[i].pw_narrow ? 6 : 5;
        DDPM_REQUIRE(variant_of[i] >= 0, "wgrad group: no kernel for nco=%d items/wave=%d", plans[i].nco, plans[i].maxi);
    }
    long long off = align16(sizeof(WgradGroupHeader));
    for (int v = 0; v < WG_VARIANTS; ++v) {
        int cnt = 0;
        for (int i = 0; i < n; ++i) cnt += variant_of[i] == v;
        h->v[v].nprob = cnt;
        h->v[v].args_off = off;
        off = align16(off + (long long)cnt * (v >= 5 ? sizeof(PwArgs) : sizeof(WgradArgs)));
        h->v[v].first_off = off;
        off = align16(off + (long long)(cnt + 1) * sizeof(int));
        WgradArgs* args = (WgradArgs*)(base + h->v[v].args_off);
        PwArgs* pargs = (PwArgs*)(base + h->v[v].args_off);
        int* first = (int*)(base + h->v[v].first_off);
        int k = 0, blocks = 0;
        long long lds = 0;
        for (int i = 0; i < n; ++i) {
            if (variant_of[i] != v) continue;
            if (v >= 5) pargs[k] = plans[i].pw; else args[k] = plans[i].a;
            first[k] = blocks;
            blocks += (int)(plans[i].grid.x * plans[i].grid.y * plans[i].grid.z);
            if ((long long)plans[i].lds > lds) lds = (long long)plans[i].lds;
            ++k;
        }
        first[k] = blocks;
        h->v[v].total_blocks = blocks;
        h->v[v].lds = lds;
    }
    return DDPM_OK;
}

template <int NCO, int MAXI>
static int launch_wgrad_group(const WgradGroupHeader::V& v, const char* dev, hipStream_t stream) {
    auto k = conv_wgrad_group_kernel<NCO, MAXI>;
    if (v.lds + sizeof(WgradArgs) + 16 > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(v.lds + sizeof(WgradArgs) + 16)) != hipSuccess)
            return ddpm_check_launch("wgrad group: raise dynamic LDS limit");
    }
    hipLaunchKernelGGL(k, dim3(v.total_blocks), dim3(256), (size_t)v.lds + ((sizeof(WgradArgs) + 15) & ~15), stream, (const WgradArgs*)(dev + v.args_off),
                       (const int*)(dev + v.first_off), v.nprob);
    return ddpm_check_launch("conv_wgrad_group");
}

extern "C" int ddpm_conv_wgrad_group_launch(const void* host_blob, const void* dev_blob, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(host_blob && dev_blob, "wgrad group: null blob");
    const WgradGroupHeader* h = (const WgradGroupHeader*)host_blob;
    DDPM_REQUIRE(h->magic == 0x5747524144LL, "wgrad group: blob was not produced by ddpm_conv_wgrad_group_plan");
    const char* dev = (const char*)dev_blob;
    for (int v = 0; v < WG_VARIANTS; ++v) {
        if (h->v[v].nprob == 0) continue;
        int rc = DDPM_OK;
        switch (v) {
            case 0: rc = launch_wgrad_group<1, 1>(h->v[v], dev, stream); break;
            case 1: rc = launch_wgrad_group<2, 1>(h->v[v], dev, stream); break;
            case 2: rc = launch_wgrad_group<4, 1>(h->v[v], dev, stream); break;
            case 3: rc = launch_wgrad_group<1, 3>(h->v[v], dev, stream); break;
            case 4: rc = launch_wgrad_group<2, 5>(h->v[v], dev, stream); break;
            case 5: {
                static bool attr = false;
                if (!attr) {
                    if (hipFuncSetAttribute((const void*)pw_wgrad_group_kernel<PW_RS>, hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS) != hipSuccess)
                        return ddpm_check_launch("wgrad group: raise dynamic LDS limit");
                    attr = true;
                }
                hipLaunchKernelGGL(pw_wgrad_group_kernel<PW_RS>, dim3(h->v[v].total_blocks), dim3(256), PW_LDS, stream,
                                   (const PwArgs*)(dev + h->v[v].args_off), (const int*)(dev + h->v[v].first_off), h->v[v].nprob);
                rc = ddpm_check_launch("pw_wgrad_group");
            } break;
            default:
                hipLaunchKernelGGL(pw_wgrad_group_kernel<PW_RS_NARROW>, dim3(h->v[v].total_blocks), dim3(256), PW_LDS_NARROW, stream,
                                   (const PwArgs*)(dev + h->v[v].args_off), (const int*)(dev + h->v[v].first_off), h->v[v].nprob);
                rc = ddpm_check_launch("pw_wgrad_group (narrow)");
                break;
        }
        if (rc) return rc;
    }
    return DDPM_OK;
}
