// The low-resolution part of the UNet as ONE launch: a workgroup carries its images through a whole program of layers
// with the activations resident in LDS.
//
// At <= 16x16 pixels an image of the denoiser is 16 KB (32 KB expanded inside an inverted-residual block): a 160 KB LDS
// holds everything a workgroup needs to run stage after stage (unets.py:127-131,230-237; constituent_blocks.py:135-147,
// 227-243; conv_block.py:132-136 of the reference) without returning to HBM and without a kernel boundary per layer.
// The host describes the work as a table of operations (ddpm_lds_op) on tiles of an LDS arena it has laid out:
//
//     LOAD / STORE   global NHWC bf16  <->  columns of a tile
//     CONV           3x3 stride 1 / 3x3 stride 2 / 1x1 / transposed 2x2 stride 2 from a tile into another tile:
//                    A fragments from LDS, weights streamed from L2 straight into MFMA B fragments (each weight element is
//                    used once per workgroup, so LDS staging would only add a hop), v_mfma_f32_32x32x16_bf16, fp32
//                    accumulation; K is split over the waves when a layer has fewer than eight 32x32 output tiles
//     GN             GroupNorm (+ activation, dropout mask, residual) in place, statistics inside the workgroup
//     DW             depthwise 3x3 in place
//     ADDT / MASK    broadcast add of a time embedding / dropout of a block input
//
// Every operation ends with a workgroup barrier; no workgroup ever reads what another one wrote.  In training the GN
// operation also writes what the backward pass needs (raw convolution outputs, statistics, normalised activations).
#include "common.h"
#include "ddpm_hip.h"

namespace {

constexpr int LT = 512;                 // threads per workgroup
constexpr int LW = LT / 64;             // waves
constexpr int ZROW = 0;                 // arena: 1056 bytes that stay zero (what out-of-image taps read)
constexpr int ZROW_BYTES = 1056;
constexpr int STAT = ZROW_BYTES;        // float [512][2]: mean, rstd per (image, group) of the running GN operation
constexpr int STAT_BYTES = 4096;
constexpr int ARENA0 = ZROW_BYTES + STAT_BYTES;
constexpr int LDS_MAX = 160 * 1024;
constexpr int RED_BYTES = LW * 32 * 33 * 4;     // K-split partial tiles

__device__ __forceinline__ int ilog2(int v) { return 31 - __builtin_clz(v); }

// ---------------------------------------------------------------------------------------------- data movement
__device__ __forceinline__ void op_load(const ddpm_lds_op& op, char* smem, int n0, int N, int ipw, int tid) {
    const int hw = op.w_out * op.w_out, R = ipw * hw, vpr = op.cout >> 3, lv = ilog2(vpr), lhw = ilog2(hw);
    const bf16* g = (const bf16*)op.p[0];
    const long long pix0 = (long long)n0 * hw;
    for (int idx = tid; idx < R * vpr; idx += LT) {
        const int row = idx >> lv, v = idx & (vpr - 1);
        bf16x8 val;
#pragma unroll
        for (int j = 0; j < 8; ++j) val[j] = (bf16)0.f;
        if (n0 + (row >> lhw) < N) val = *(const bf16x8*)(g + (pix0 + row) * op.gstride + op.gcol0 + v * 8);
        *(bf16x8*)(smem + op.dst_off + row * op.dst_stride + (op.col0 + v * 8) * 2) = val;
    }
}

__device__ __forceinline__ void op_store(const ddpm_lds_op& op, const char* smem, int n0, int N, int ipw, int tid) {
    const int hw = op.w_out * op.w_out, R = ipw * hw, vpr = op.cout >> 3, lv = ilog2(vpr), lhw = ilog2(hw);
    bf16* g = (bf16*)op.p[0];
    const long long pix0 = (long long)n0 * hw;
    for (int idx = tid; idx < R * vpr; idx += LT) {
        const int row = idx >> lv, v = idx & (vpr - 1);
        if (n0 + (row >> lhw) < N)
            *(bf16x8*)(g + (pix0 + row) * op.gstride + op.gcol0 + v * 8) = *(const bf16x8*)(smem + op.src_off + row * op.src_stride + (op.col0 + v * 8) * 2);
    }
}

// tile[row][c] += temb[image][c]  (the stage-input broadcast add, unets.py:128,237), rounded to bf16 as the convolution
// loaders of the per-layer kernels round it
__device__ __forceinline__ void op_addt(const ddpm_lds_op& op, char* smem, int n0, int N, int ipw, int tid) {
    const int hw = op.w_out * op.w_out, R = ipw * hw, C = op.cout, vpr = C >> 3, lv = ilog2(vpr), lhw = ilog2(hw);
    const float* temb = (const float*)op.p[0];
    for (int idx = tid; idx < R * vpr; idx += LT) {
        const int row = idx >> lv, v = idx & (vpr - 1);
        const int n = n0 + (row >> lhw);
        if (n >= N) continue;
        char* p = smem + op.dst_off + row * op.dst_stride + (op.col0 + v * 8) * 2;
        bf16x8 x = *(bf16x8*)p;
        const f32x4 a0 = *(const f32x4*)(temb + (long long)n * C + v * 8), a1 = *(const f32x4*)(temb + (long long)n * C + v * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[j] = (bf16)((float)x[j] + a0[j]); x[4 + j] = (bf16)((float)x[4 + j] + a1[j]); }
        *(bf16x8*)p = x;
    }
}

// dst = src * keep-bit * scale: the dropout on the input of an inverted-residual block (conv_block.py:57-58); the residual
// keeps reading the clean source tile
__device__ __forceinline__ void op_mask(const ddpm_lds_op& op, char* smem, int n0, int N, int ipw, int tid) {
    const int hw = op.w_out * op.w_out, R = ipw * hw, C = op.cout, vpr = C >> 3, lv = ilog2(vpr), lhw = ilog2(hw);
    const unsigned char* mk = (const unsigned char*)op.p[0];
    const long long pix0 = (long long)n0 * hw;
    for (int idx = tid; idx < R * vpr; idx += LT) {
        const int row = idx >> lv, v = idx & (vpr - 1);
        bf16x8 x = *(const bf16x8*)(smem + op.src_off + row * op.src_stride + v * 16);
        if (mk) {                                      // (no keep bits: a plain copy between tiles)
            unsigned m = 0;
            if (n0 + (row >> lhw) < N) m = mk[((pix0 + row) * C + v * 8) >> 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (m >> j) & 1 ? (bf16)((float)x[j] * op.mscale) : (bf16)0.f;
        }
        *(bf16x8*)(smem + op.dst_off + row * op.dst_stride + v * 16) = x;
    }
}

// ---------------------------------------------------------------------------------------------- depthwise 3x3, in place
constexpr int DW_NV = 8;                 // vectors per thread the accumulators are sized for (host-checked)

__device__ __forceinline__ void op_dw(const ddpm_lds_op& op, char* smem, int ipw, int tid) {
    const int W = op.w_out, lw = ilog2(W), hw = W * W, R = ipw * hw, C = op.cout, vpr = C >> 3, lv = ilog2(vpr);
    const int v = tid & (vpr - 1), r0 = tid >> lv, rpp = LT >> lv;
    const int nv = (R + rpp - 1) / rpp;
    const float* wd = (const float*)op.p[0];
    const float* bd = (const float*)op.p[1];
    char* tile = smem + op.dst_off;
    const int st = op.dst_stride;
    float acc[DW_NV][8];
    {
        const f32x4 b0 = *(const f32x4*)(bd + v * 8), b1 = *(const f32x4*)(bd + v * 8 + 4);
#pragma unroll
        for (int k = 0; k < DW_NV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[k][j] = b0[j]; acc[k][4 + j] = b1[j]; }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        float w[8];
        const f32x4 w0 = *(const f32x4*)(wd + tap * C + v * 8), w1 = *(const f32x4*)(wd + tap * C + v * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = w0[j]; w[4 + j] = w1[j]; }
#pragma unroll
        for (int k = 0; k < DW_NV; ++k) {
            if (k < nv) {
                const int row = r0 + k * rpp;
                const int p = row & (hw - 1), ny = (p >> lw) + dy, nx = (p & (W - 1)) + dx;
                if (row < R && ny >= 0 && ny < W && nx >= 0 && nx < W) {
                    const bf16x8 x = *(const bf16x8*)(tile + (row + dy * W + dx) * st + v * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[k][j] += (float)x[j] * w[j];
                }
            }
        }
    }
    __syncthreads();                                   // every neighbour has been read: the tile may be overwritten
#pragma unroll
    for (int k = 0; k < DW_NV; ++k) {
        const int row = r0 + k * rpp;
        if (k < nv && row < R) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[k][j];
            *(bf16x8*)(tile + row * st + v * 16) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------- GroupNorm (+ act, mask, residual), in place
// Statistics over the bf16 values the tile holds (what a separate GroupNorm launch would read back), fp32 sums: the
// threads of one (image, group) pair walk its pixels in 4/8/16-byte pieces and meet by shuffles.
template <int ACT>
__device__ __forceinline__ void op_gn(const ddpm_lds_op& op, char* smem, int n0, int N, int ipw, int tid) {
    const int W = op.w_out, hw = W * W, lhw = ilog2(hw), R = ipw * hw, C = op.cout, G = op.groups, cpg = C / G, lcpg = ilog2(cpg);
    float* s_stat = (float*)(smem + STAT);
    char* tile = smem + op.dst_off + op.col0 * 2;
    const int st = op.dst_stride;
    {
        const int pairs = ipw * G, tpp = LT / pairs, ltpp = ilog2(tpp);          // host-checked: 8 <= pairs <= 512, power of two
        const int pair = tid >> ltpp, sub = tid & (tpp - 1);
        const int img = pair / G, g = pair - img * G;
        const int pb = cpg * 2, vb = pb < 16 ? pb : 16, ppp = pb / vb, lppp = ilog2(ppp);
        const int pieces = hw * ppp;
        const char* base = tile + (img * hw) * st + g * pb;
        float s = 0.f, ss = 0.f;
        for (int q = sub; q < pieces; q += tpp) {
            const char* a = base + (q >> lppp) * st + (q & (ppp - 1)) * vb;
            if (vb == 16) {
                const bf16x8 x = *(const bf16x8*)a;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)x[j]; s += f; ss += f * f; }
            } else if (vb == 8) {
                const bf16x4 x = *(const bf16x4*)a;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float f = (float)x[j]; s += f; ss += f * f; }
            } else {
                const float f0 = (float)*(const bf16*)a, f1 = (float)*(const bf16*)(a + 2);
                s += f0 + f1; ss += f0 * f0 + f1 * f1;
            }
        }
        for (int o = 1; o < tpp; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (sub == 0) {
            const float im = 1.0f / (float)(cpg * hw);
            const float mean = s * im;
            const float rstd = rsqrtf(fmaxf(ss * im - mean * mean, 0.f) + op.eps);
            s_stat[pair * 2] = mean; s_stat[pair * 2 + 1] = rstd;
            float* so = (float*)op.p[5];
            if (so && n0 + img < N) { so[((long long)(n0 + img) * G + g) * 2] = mean; so[((long long)(n0 + img) * G + g) * 2 + 1] = rstd; }
        }
    }
    __syncthreads();
    const int vpr = C >> 3, lv = ilog2(vpr), v = tid & (vpr - 1), r0 = tid >> lv, rpp = LT >> lv;
    float ga[8], be[8];
    {
        const float* gp = (const float*)op.p[0] + v * 8;
        const float* bp = (const float*)op.p[1] + v * 8;
        const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4), b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { ga[j] = g0[j]; ga[4 + j] = g1[j]; be[j] = b0[j]; be[4 + j] = b1[j]; }
    }
    const unsigned char* mk = (const unsigned char*)op.p[2];
    bf16* save_raw = (bf16*)op.p[3];
    bf16* save_y = (bf16*)op.p[4];
    const bool has_res = op.aux_stride != 0;
    const bool act_on = op.act != 0;
    const long long pix0 = (long long)n0 * hw;
    for (int row = r0; row < R; row += rpp) {
        const int img = row >> lhw;
        const bool valid = n0 + img < N;
        char* p = tile + row * st + v * 16;
        const bf16x8 x = *(const bf16x8*)p;
        const long long e = (pix0 + row) * C + v * 8;
        if (save_raw && valid) *(bf16x8*)(save_raw + e) = x;
        unsigned m = 0xFFu;
        if (mk && valid) m = mk[e >> 3];
        bf16x8 rv;
        if (has_res) rv = *(const bf16x8*)(smem + op.aux_off + row * op.aux_stride + v * 16);
        bf16x8 o;
        const float* sp = s_stat + (img * G) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (v * 8 + j) >> lcpg;
            const float mean = sp[g * 2], rstd = sp[g * 2 + 1];
            const float sc = ga[j] * rstd, sh = be[j] - mean * sc;
            float y = (float)x[j] * sc + sh;
            if (act_on) {
                if constexpr (ACT == ACT_GELU) y = gelu_fwd(y);
                else if constexpr (ACT == ACT_SILU) y = y * sigmoid_fast(y);
                else y = y > 0.f ? y : 0.f;
            }
            if (mk) y = (m >> j) & 1 ? y * op.mscale : 0.f;
            if (has_res) y += (float)rv[j];
            o[j] = (bf16)y;
        }
        *(bf16x8*)p = o;
        if (save_y && valid) *(bf16x8*)(save_y + e) = o;
    }
}

// ---------------------------------------------------------------------------------------------- convolution
struct ConvGeo {
    int kind, src_off, src_stride, w_in, lw_out, w_out_mask, hw_in;
};

// Byte address in LDS of the A fragment row a lane reads for tap `tap` (without the channel offset of the k-step):
// the input pixel of the lane's output row under that tap, or the zero row outside the image.
__device__ __forceinline__ int a_base(const ConvGeo& g, int tap, bool m_ok, int img, int p, int hoff) {
    int row;
    bool ok = m_ok;
    if (g.kind == DDPM_CONV_3X3_S1 || g.kind == DDPM_CONV_3X3_S2) {
        const int s = g.kind == DDPM_CONV_3X3_S2 ? 2 : 1;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int iy = (p >> g.lw_out) * s + dy, ix = (p & g.w_out_mask) * s + dx;
        ok = ok && iy >= 0 && iy < g.w_in && ix >= 0 && ix < g.w_in;
        row = img * g.hw_in + iy * g.w_in + ix;
    } else {
        row = img * g.hw_in + p;                       // 1x1, and one parity class of the transposed convolution
    }
    return (ok ? g.src_off + row * g.src_stride : ZROW) + hoff;
}

// One 32x32 output tile's accumulation over the k-steps [8 c_lo, 8 c_hi) of the flat (tap, 16-channel step) sequence.
// KPT = k-steps per tap (Cin / 16).  Weight fragments are double-buffered in registers, eight k-steps per buffer: the
// loads of the next eight are in flight while the current eight multiply.
template <int KPT>
__device__ __forceinline__ f32x16 conv_acc(const char* smem, const ConvGeo& g, int ntaps, bool m_ok, int img, int p, int lane,
                                           const bf16* wlane, long long slice, int c_lo, int c_hi) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int hoff = (lane >> 5) * 16;
    bf16x8 b0[8], b1[8];
    constexpr int TPC = KPT >= 8 ? 1 : 8 / KPT;        // taps per chunk
    auto load_b = [&](bf16x8 (&b)[8], int c) {
        if constexpr (KPT >= 8) {
            const int s0 = c * 8, tap = s0 / KPT, kc0 = s0 - tap * KPT;
            const bf16* wp = wlane + tap * slice + kc0 * 16;
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = *(const bf16x8*)(wp + u * 16);
        } else {
#pragma unroll
            for (int tt = 0; tt < TPC; ++tt) {
                const int tap = c * TPC + tt;
                if (tap < ntaps) {
                    const bf16* wp = wlane + tap * slice;
#pragma unroll
                    for (int kk = 0; kk < KPT; ++kk) b[tt * KPT + kk] = *(const bf16x8*)(wp + kk * 16);
                }
            }
        }
    };
    auto compute = [&](bf16x8 (&b)[8], int c) {
        if constexpr (KPT >= 8) {
            const int s0 = c * 8, tap = s0 / KPT, kc0 = s0 - tap * KPT;
            const char* ap = smem + a_base(g, tap, m_ok, img, p, hoff) + kc0 * 32;
            bf16x8 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = *(const bf16x8*)(ap + u * 32);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int tt = 0; tt < TPC; ++tt) {
                const int tap = c * TPC + tt;
                if (tap < ntaps) {
                    const char* ap = smem + a_base(g, tap, m_ok, img, p, hoff);
#pragma unroll
                    for (int kk = 0; kk < KPT; ++kk) {
                        const bf16x8 a = *(const bf16x8*)(ap + kk * 32);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[tt * KPT + kk], acc, 0, 0, 0);
                    }
                }
            }
        }
    };
    int c = c_lo;
    if (c < c_hi) load_b(b0, c);
    for (; c < c_hi; c += 2) {
        if (c + 1 < c_hi) load_b(b1, c + 1);
        compute(b0, c);
        if (c + 1 < c_hi) {
            if (c + 2 < c_hi) load_b(b0, c + 2);
            compute(b1, c + 1);
        }
    }
    return acc;
}

__device__ __forceinline__ f32x16 conv_acc_any(int kpt, const char* smem, const ConvGeo& g, int ntaps, bool m_ok, int img, int p, int lane,
                                               const bf16* wlane, long long slice, int c_lo, int c_hi) {
    switch (kpt) {
        case 1: return conv_acc<1>(smem, g, ntaps, m_ok, img, p, lane, wlane, slice, c_lo, c_hi);
        case 2: return conv_acc<2>(smem, g, ntaps, m_ok, img, p, lane, wlane, slice, c_lo, c_hi);
        case 4: return conv_acc<4>(smem, g, ntaps, m_ok, img, p, lane, wlane, slice, c_lo, c_hi);
        case 8: return conv_acc<8>(smem, g, ntaps, m_ok, img, p, lane, wlane, slice, c_lo, c_hi);
        case 16: return conv_acc<16>(smem, g, ntaps, m_ok, img, p, lane, wlane, slice, c_lo, c_hi);
        default: return conv_acc<32>(smem, g, ntaps, m_ok, img, p, lane, wlane, slice, c_lo, c_hi);
    }
}

// Row of the destination tile that receives row `mi` of an M tile of parity class `cls` (-1: not a transposed convolution).
__device__ __forceinline__ int out_row(int mi, int cls, int w_in, int lw_in, int lhw_in) {
    if (cls < 0) return mi;
    const int img = mi >> lhw_in, p = mi & ((1 << lhw_in) - 1);
    const int iy = p >> lw_in, ix = p & (w_in - 1);
    return (img << (lhw_in + 2)) + ((2 * iy + (cls >> 1)) << (lw_in + 1)) + 2 * ix + (cls & 1);
}

__device__ __forceinline__ void op_conv(const ddpm_lds_op& op, char* smem, int ipw, int tid) {
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const bool convt = op.kind == DDPM_CONVT_2X2_S2;
    const int Cin = op.cin, Cout = op.cout, kpt = Cin >> 4;
    const int ntaps = (op.kind == DDPM_CONV_3X3_S1 || op.kind == DDPM_CONV_3X3_S2) ? 9 : 1;
    const int hw_in = op.w_in * op.w_in, hw_out = op.w_out * op.w_out;
    const int Rm = ipw * (convt ? hw_in : hw_out);     // rows of the GEMM (per parity class for the transposed convolution)
    const int lw_m = ilog2(convt ? op.w_in : op.w_out), lhw_m = 2 * lw_m;
    const int MT = (Rm + 31) >> 5, NT = Cout >> 5, ncls = convt ? 4 : 1;
    const int T = ncls * MT * NT;
    const int S = ntaps * kpt, NCH = (S + 7) >> 3;
    const long long slice = (long long)op.gstride * Cin;          // gstride: weight rows per tap slice (Cout rounded up to 32)
    const bf16* w = (const bf16*)op.p[0];
    const float* bias = (const float*)op.p[1];
    ConvGeo g;
    g.kind = op.kind; g.src_off = op.src_off; g.src_stride = op.src_stride; g.w_in = op.w_in;
    g.lw_out = lw_m; g.w_out_mask = (1 << lw_m) - 1; g.hw_in = hw_in;
    char* dst = smem + op.dst_off + op.col0 * 2;
    const int dstr = op.dst_stride;
    const int Rout = ipw * hw_out;

    auto tile_of = [&](int t, int& mt, int& nt, int& cls) {
        mt = t % MT;
        const int q = t / MT;
        nt = q % NT;
        cls = convt ? q / NT : -1;
    };
    if (T >= LW) {
        for (int t = wave; t < T; t += LW) {
            int mt, nt, cls;
            tile_of(t, mt, nt, cls);
            const int m = mt * 32 + r;
            const bf16* wlane = w + (cls > 0 ? cls * slice : 0) + (long long)(nt * 32 + r) * Cin + h * 8;
            const f32x16 acc = conv_acc_any(kpt, smem, g, ntaps, m < Rm, m >> lhw_m, m & ((1 << lhw_m) - 1), lane, wlane, slice, 0, NCH);
            const float b = bias ? bias[nt * 32 + r] : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int mi = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (mi < Rm) {
                    const int orow = out_row(mi, cls, op.w_in, lw_m, lhw_m);
                    *(bf16*)(dst + orow * dstr + (nt * 32 + r) * 2) = (bf16)(acc[i] + b);
                }
            }
        }
        return;
    }
    // fewer tiles than waves: LW / T waves share a tile, each with a contiguous part of the k-steps; fp32 partial tiles
    // meet in LDS and are added in a fixed order
    const int ks = LW / T;                              // host-checked: T is a power of two
    const int t = wave % T, part = wave / T;
    float* s_red = (float*)(smem + op.aux_off);         // [LW][32][33]
    {
        int mt, nt, cls;
        tile_of(t, mt, nt, cls);
        const int m = mt * 32 + r;
        const bf16* wlane = w + (cls > 0 ? cls * slice : 0) + (long long)(nt * 32 + r) * Cin + h * 8;
        const int c_lo = part * NCH / ks, c_hi = (part + 1) * NCH / ks;
        const f32x16 acc = conv_acc_any(kpt, smem, g, ntaps, m < Rm, m >> lhw_m, m & ((1 << lhw_m) - 1), lane, wlane, slice, c_lo, c_hi);
#pragma unroll
        for (int i = 0; i < 16; ++i) s_red[(wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * h) * 33 + r] = acc[i];
    }
    __syncthreads();
    for (int e = tid; e < T * 1024; e += LT) {
        const int tt = e >> 10, rr = (e >> 5) & 31, cc = e & 31;
        int mt, nt, cls;
        tile_of(tt, mt, nt, cls);
        const int mi = mt * 32 + rr;
        if (mi >= Rm) continue;
        float v = bias ? bias[nt * 32 + cc] : 0.f;
        for (int pp = 0; pp < ks; ++pp) v += s_red[((pp * T + tt) * 32 + rr) * 33 + cc];
        const int orow = out_row(mi, cls, op.w_in, lw_m, lhw_m);
        *(bf16*)(dst + orow * dstr + (nt * 32 + cc) * 2) = (bf16)v;
    }
    (void)Rout;
}

template <int ACT>
__global__ __launch_bounds__(LT) void unet_lds_kernel(const ddpm_lds_op* __restrict__ ops, int N, int ipw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * ipw;
    for (int i = tid; i < ZROW_BYTES / 4; i += LT) ((float*)smem)[i] = 0.f;
    __syncthreads();
    for (int pc = 0;; ++pc) {
        const ddpm_lds_op& op = ops[pc];
        const int type = op.type;
        if (type == DDPM_LOP_END) break;
        switch (type) {
            case DDPM_LOP_LOAD: op_load(op, smem, n0, N, ipw, tid); break;
            case DDPM_LOP_STORE: op_store(op, smem, n0, N, ipw, tid); break;
            case DDPM_LOP_CONV: op_conv(op, smem, ipw, tid); break;
            case DDPM_LOP_GN: op_gn<ACT>(op, smem, n0, N, ipw, tid); break;
            case DDPM_LOP_DW: op_dw(op, smem, ipw, tid); break;
            case DDPM_LOP_ADDT: op_addt(op, smem, n0, N, ipw, tid); break;
            case DDPM_LOP_MASK: op_mask(op, smem, n0, N, ipw, tid); break;
            default: break;
        }
        __syncthreads();
    }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int ddpm_unet_lds_arena0(void) { return ARENA0; }
extern "C" int ddpm_unet_lds_reduce_bytes(void) { return RED_BYTES; }
extern "C" int ddpm_unet_lds_max_bytes(void) { return LDS_MAX; }

// Checks every operation of the program against what the kernel assumes and returns the LDS bytes the program needs
// (or a negative error code).  `ops` is the HOST copy of the table.
extern "C" long long ddpm_unet_lds_check(const ddpm_lds_op* ops, int nops, int ipw) {
    DDPM_REQUIRE(ops && nops > 0, "unet_lds: empty program");
    DDPM_REQUIRE(ipw == 1 || ipw == 2 || ipw == 4 || ipw == 8, "unet_lds: images per workgroup must be 1, 2, 4 or 8 (got %d)", ipw);
    DDPM_REQUIRE(ops[nops - 1].type == DDPM_LOP_END, "unet_lds: the program must end with DDPM_LOP_END");
    long long need = ARENA0;
    auto extent = [&](int off, int stride, int rows, int col0, int cols) -> bool {
        if (off < ARENA0 || (off & 15) || stride <= 0 || (stride & 15) || (col0 + cols) * 2 > stride) return false;
        const long long end = (long long)off + (long long)rows * stride;
        if (end > need) need = end;
        return true;
    };
    for (int i = 0; i < nops - 1; ++i) {
        const ddpm_lds_op& o = ops[i];
        const int hw_o = o.w_out * o.w_out, hw_i = o.w_in * o.w_in;
        DDPM_REQUIRE(o.type > DDPM_LOP_END && o.type <= DDPM_LOP_MASK, "unet_lds: op %d has unknown type %d", i, o.type);
        DDPM_REQUIRE(pow2(o.w_out) && o.w_out <= 64, "unet_lds: op %d: image width %d must be a power of two <= 64", i, o.w_out);
        DDPM_REQUIRE(pow2(o.cout) && o.cout >= 8 && o.cout <= 1024, "unet_lds: op %d: channel count %d must be a power of two in [8, 1024]", i, o.cout);
        DDPM_REQUIRE(o.col0 >= 0 && (o.col0 & 7) == 0, "unet_lds: op %d: first column must be a multiple of 8", i);
        switch (o.type) {
            case DDPM_LOP_LOAD:
            case DDPM_LOP_STORE:
                DDPM_REQUIRE(o.p[0] && (o.gstride & 7) == 0 && (o.gcol0 & 7) == 0 && o.gcol0 + o.cout <= o.gstride, "unet_lds: op %d: bad global layout", i);
                if (o.type == DDPM_LOP_LOAD) DDPM_REQUIRE(extent(o.dst_off, o.dst_stride, ipw * hw_o, o.col0, o.cout), "unet_lds: op %d: bad tile", i);
                else DDPM_REQUIRE(extent(o.src_off, o.src_stride, ipw * hw_o, o.col0, o.cout), "unet_lds: op %d: bad tile", i);
                break;
            case DDPM_LOP_ADDT:
                DDPM_REQUIRE(o.p[0], "unet_lds: op %d: no embedding", i);
                DDPM_REQUIRE(extent(o.dst_off, o.dst_stride, ipw * hw_o, o.col0, o.cout), "unet_lds: op %d: bad tile", i);
                break;
            case DDPM_LOP_MASK:
                DDPM_REQUIRE(o.col0 == 0, "unet_lds: op %d: column offset", i);
                DDPM_REQUIRE(extent(o.dst_off, o.dst_stride, ipw * hw_o, 0, o.cout) && extent(o.src_off, o.src_stride, ipw * hw_o, 0, o.cout),
                             "unet_lds: op %d: bad tile", i);
                break;
            case DDPM_LOP_DW: {
                DDPM_REQUIRE(o.p[0] && o.p[1] && o.col0 == 0, "unet_lds: op %d: depthwise needs weights and bias", i);
                DDPM_REQUIRE(extent(o.dst_off, o.dst_stride, ipw * hw_o, 0, o.cout), "unet_lds: op %d: bad tile", i);
                const int vpr = o.cout / 8;
                DDPM_REQUIRE(vpr <= LT, "unet_lds: op %d: too many channels", i);
                const int rpp = LT / vpr;
                DDPM_REQUIRE((ipw * hw_o + rpp - 1) / rpp <= DW_NV, "unet_lds: op %d: depthwise tile of %d x %d does not fit the accumulators", i, ipw * hw_o, o.cout);
                break;
            }
            case DDPM_LOP_GN: {
                DDPM_REQUIRE(o.p[0] && o.p[1], "unet_lds: op %d: GroupNorm needs gamma and beta", i);
                DDPM_REQUIRE(o.groups > 0 && o.cout % o.groups == 0 && pow2(o.groups), "unet_lds: op %d: bad group count %d", i, o.groups);
                const int cpg = o.cout / o.groups, pairs = ipw * o.groups;
                DDPM_REQUIRE(cpg >= 2 && pairs >= 8 && pairs <= LT, "unet_lds: op %d: %d channels per group, %d (image, group) pairs", i, cpg, pairs);
                DDPM_REQUIRE(o.cout / 8 <= LT, "unet_lds: op %d: too many channels", i);
                DDPM_REQUIRE(extent(o.dst_off, o.dst_stride, ipw * hw_o, o.col0, o.cout), "unet_lds: op %d: bad tile", i);
                if (o.aux_stride) DDPM_REQUIRE(extent(o.aux_off, o.aux_stride, ipw * hw_o, 0, o.cout), "unet_lds: op %d: bad residual tile", i);
                DDPM_REQUIRE(o.act == DDPM_ACT_NONE || o.act == DDPM_ACT_GELU || o.act == DDPM_ACT_SILU || o.act == DDPM_ACT_RELU,
                             "unet_lds: op %d: activation %d is not built into this kernel", i, o.act);
                break;
            }
            case DDPM_LOP_CONV: {
                DDPM_REQUIRE(o.p[0], "unet_lds: op %d: convolution without weights", i);
                DDPM_REQUIRE(o.kind == DDPM_CONV_3X3_S1 || o.kind == DDPM_CONV_3X3_S2 || o.kind == DDPM_CONV_1X1 || o.kind == DDPM_CONVT_2X2_S2,
                             "unet_lds: op %d: convolution kind %d", i, o.kind);
                DDPM_REQUIRE(pow2(o.w_in) && pow2(o.cin) && o.cin >= 16 && o.cin <= 512 && o.cout >= 32, "unet_lds: op %d: Cin %d / Cout %d / width %d", i, o.cin,
                             o.cout, o.w_in);
                const int want_in = o.kind == DDPM_CONV_3X3_S2 ? 2 * o.w_out : o.kind == DDPM_CONVT_2X2_S2 ? o.w_out / 2 : o.w_out;
                DDPM_REQUIRE(o.w_in == want_in, "unet_lds: op %d: input width %d does not match output width %d", i, o.w_in, o.w_out);
                DDPM_REQUIRE(o.gstride >= o.cout && (o.gstride & 31) == 0, "unet_lds: op %d: weight rows per tap %d", i, o.gstride);
                DDPM_REQUIRE(extent(o.src_off, o.src_stride, ipw * hw_i, 0, o.cin), "unet_lds: op %d: bad source tile", i);
                DDPM_REQUIRE(extent(o.dst_off, o.dst_stride, ipw * hw_o, o.col0, o.cout), "unet_lds: op %d: bad destination tile", i);
                const long long s0 = o.src_off, s1 = s0 + (long long)ipw * hw_i * o.src_stride, d0 = o.dst_off, d1 = d0 + (long long)ipw * hw_o * o.dst_stride;
                DDPM_REQUIRE(s1 <= d0 || d1 <= s0, "unet_lds: op %d: source and destination tiles overlap", i);
                const int rm = ipw * (o.kind == DDPM_CONVT_2X2_S2 ? hw_i : hw_o);
                const int T = (o.kind == DDPM_CONVT_2X2_S2 ? 4 : 1) * ((rm + 31) / 32) * (o.cout / 32);
                if (T < LW) {
                    DDPM_REQUIRE(pow2(T), "unet_lds: op %d: %d output tiles", i, T);
                    DDPM_REQUIRE(o.aux_off >= ARENA0 && (o.aux_off & 15) == 0, "unet_lds: op %d: no reduction scratch", i);
                    const long long a0 = o.aux_off, a1 = a0 + RED_BYTES;
                    DDPM_REQUIRE((a1 <= s0 || s1 <= a0) && (a1 <= d0 || d1 <= a0), "unet_lds: op %d: reduction scratch overlaps a tile", i);
                    if (a1 > need) need = a1;
                }
                break;
            }
        }
    }
    DDPM_REQUIRE(need <= LDS_MAX, "unet_lds: the program needs %lld bytes of LDS (limit %d)", need, LDS_MAX);
    return need;
}

// Runs the program: ops_dev = the table in device memory, ops_host = the same table in host memory (validated here before
// anything is launched).  One workgroup per `ipw` images.
extern "C" int ddpm_unet_lds_run(const ddpm_lds_op* ops_dev, const ddpm_lds_op* ops_host, int nops, int N, int ipw, int act,
                                 ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(ops_dev && N > 0, "unet_lds: null table or empty batch");
    const long long need = ddpm_unet_lds_check(ops_host, nops, ipw);
    if (need < 0) return (int)need;
    DDPM_REQUIRE(act == ACT_GELU || act == ACT_SILU || act == ACT_RELU, "unet_lds: activation %d is not built into this kernel", act);
    void (*k)(const ddpm_lds_op*, int, int) = act == ACT_GELU ? unet_lds_kernel<ACT_GELU> : act == ACT_SILU ? unet_lds_kernel<ACT_SILU>
                                                                                                             : unet_lds_kernel<ACT_RELU>;
    static int raised[3] = {0, 0, 0};
    const int slot = act == ACT_GELU ? 0 : act == ACT_SILU ? 1 : 2;
    if (need > raised[slot]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) {
            ddpm_set_error("unet_lds: cannot reserve %d bytes of LDS", LDS_MAX);
            return DDPM_EHIP;
        }
        raised[slot] = LDS_MAX;
    }
    hipLaunchKernelGGL(k, dim3((N + ipw - 1) / ipw), dim3(LT), (size_t)need, stream, ops_dev, N, ipw);
    return ddpm_check_launch("unet_lds");
}
