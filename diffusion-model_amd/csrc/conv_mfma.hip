// Implicit-GEMM convolution on bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for gfx950.
//
// One kernel serves the forward pass AND the data-gradient of every dense convolution in the UNet
// (3x3 stride 1/2, 1x1 / Linear, transposed 2x2 stride 2); see conv_geom.h for the tap-list form.
//
//   GEMM view per class:  M = positions (N*GH*GW),  N = output channels,  K = taps x input channels.
//   A operand  = input patch rows (one pixel = one row of 32 channels) read from LDS at pixel + tap offset
//   B operand  = weight rows [out channel][32 in channels] of the tap's slice, read from LDS
//   Wave tile  = 32 positions x (32*NT) output channels; workgroup = WAVES waves stacked along M.
//
// Per 32-channel chunk the patch (with halo) is loaded from HBM exactly once, through registers, which is
// where the fused input transforms happen: two-source channel concat (decoder skip connection), the
// per-(n, c) time-embedding broadcast add, and the input dropout mask.  Epilogue: bias, activation,
// dropout mask, residual add, staged through LDS so that global stores are 16-byte NHWC vectors.
#include "conv_geom.h"
#include "ddpm_hip.h"

int ddpm_conv_small_try(const ddpm_conv_desc* d, hipStream_t stream);   // conv_small.hip
int ddpm_conv_thin_try(const ddpm_conv_desc* d, hipStream_t stream);    // conv_thin.hip
int ddpm_conv_img_try(const ddpm_conv_desc* d, hipStream_t stream, int plain_too);   // conv_img.hip
int ddpm_conv_pipe_try(const ddpm_conv_desc* d, hipStream_t stream);   // conv_pipe.hip
int ddpm_conv_pw_stream_try(const ddpm_conv_desc* d, hipStream_t stream);   // conv_pw_stream.hip
int ddpm_conv_pw_stream_tiles(const ddpm_conv_desc* d);

struct ConvArgs {
    ConvGeom g;
    const bf16* in0; const bf16* in1; int C0, C1;
    const float* in_add;             // [N][C0] fp32, added to in-bounds pixels of source 0 (nullable)
    const unsigned char* in_mask;    // keep bits for source 0, one byte per 8 consecutive elements (nullable)
    float in_mask_scale;
    const bf16* w; const float* bias;
    void* out; const bf16* res; const unsigned char* out_mask; float out_mask_scale;
    int Cout, CoutW, out_cs, w_rows; // real output channels; rows computed; channel stride of out; rows per weight slice
    int act, out_mode;
    float* sum_out;                  // optional fp32 [N][Cout] per-image sums of the activated output (atomics)
    bf16* out2; int act2;            // optional second result act2(out), same layout: a unit without normalisation keeps its
                                     // pre-activation for the backward pass and hands the activated tensor on
    const float* in_gn; int in_gn_act; // optional (PRE kernels): in0 is a raw tensor whose GroupNorm + activation is applied on load:
                                     // fp32 [N][C0][2] = (scale, shift) per image and channel, then the activation
    float* gn_part; int gn_G, gn_PT; // optional (STATS kernels): per (image, group, output tile) sums of the rounded result and of
                                     // its squares, fp32 [N][gn_G][gn_PT][2]: what the GroupNorm that follows needs, without a
                                     // statistics pass over the tensor
};

// Loads are issued in batches of STAGE_U per thread before any of them is consumed, so that each thread keeps
// several 16-byte requests in flight (the tensors are mostly L2 / Infinity-Cache resident: latency, not bandwidth,
// is what a one-load-at-a-time loop would pay).
#define STAGE_U 4

// Diagnostic build only (-DCONV_STAMPS): lane 0 of every wave records the shader clock at phase boundaries.
#ifdef CONV_STAMPS
__device__ unsigned long long* conv_dbg = nullptr;
extern "C" int ddpm_conv_set_dbg(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(conv_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#define CONV_STAMP(i)                                                                                                        \
    do {                                                                                                                     \
        if (conv_dbg && (threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z == 0)                                       \
            conv_dbg[((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CONV_STAMP(i)
#endif

template <int WM, int WN, int NT, bool STATS = false, bool PRE = false>
__global__ __launch_bounds__(256, NT >= 4 ? 2 : 3) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int T = 256, BM = 32 * WM, BNW = 32 * NT, BN = BNW * WN;
    constexpr int OS = BN * 2 + 16;   // epilogue LDS row stride (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_m = (tid >> 6) % WM, wave_n = (tid >> 6) / WM;
    const int wave = wave_m;          // row group of this wave inside the tile
    const int cls = blockIdx.z;
    const int tap0 = g.cls_tap_begin[cls], ntaps = g.cls_tap_begin[cls + 1] - tap0;
    const int co0 = blockIdx.y * BN;
    int tile = blockIdx.x;
    const int twi = tile % g.tiles_w; tile /= g.tiles_w;
    const int thi = tile % g.tiles_h;
    const int tni = tile / g.tiles_h;
    const int n0 = tni * g.NB, oy = thi * g.TH, ox = twi * g.TW;
    const int PHW = g.PH * g.PW, npix = g.NB * PHW;
    const int Cin = a.C0 + a.C1;

    CONV_STAMP(0);
    int* s_pl = (int*)smem;                                      // pixel linear index (or -1) per patch pixel
    int* s_tap = (int*)(smem + ((npix * 4 + 15) & ~15));         // [16] weight slice of each tap of this class
    char* s_patch = (char*)(s_tap + CONV_MAX_TAPS);
    char* s_w = s_patch + npix * CONV_RS;

    // ---- patch pixel table: patch slot -> linear input pixel (n*IH + iy)*IW + ix, or -1 if padding
    for (int p = tid; p < npix; p += T) {
        const int il = div_fast(p, g.phw_magic), r = p - il * PHW;
        const int py = div_fast(r, g.pw_magic), px = r - py * g.PW;
        const int n = n0 + il, iy = oy * g.SI + g.dmin_h + py, ix = ox * g.SI + g.dmin_w + px;
        const bool ok = n < g.N && iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW;
        s_pl[p] = ok ? (n * g.IH + iy) * g.IW + ix : -1;
    }
    // tap tables: the weight slices go to LDS (looked up per staged weight row), the patch offsets stay in a register,
    // lane j holding tap j's (the MFMA loop reads them with v_readlane: a scalar load per tap would sit on its critical path)
    if (tid < CONV_MAX_TAPS) s_tap[tid] = tid < ntaps ? g.tap_slice[tap0 + tid] : 0;
    const int my_poff = lane < ntaps ? g.tap_poff[tap0 + lane] : 0;

    // ---- this lane's A row (position) -> byte offset of its pixel inside the patch
    const int thw = g.TH * g.TW;
    const int pidx = wave * 32 + (lane & 31);
    const int il = pidx >> g.thw_shift, rem = pidx - (il << g.thw_shift);
    const int ty = rem >> g.tw_shift, tx = rem - (ty << g.tw_shift);
    const int a_base = (il * PHW + ty * g.SI * g.PW + tx * g.SI) * CONV_RS + (lane >> 5) * 16;
    const int b_base = (wave_n * BNW + (lane & 31)) * CONV_RS + (lane >> 5) * 16;

    f32x16 acc[NT];
    float bias_r[NT];                 // this lane's output-channel biases, fetched now so the epilogue does not wait for them
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int co = co0 + wave_n * BNW + i * 32 + (lane & 31);
        bias_r[i] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }

    for (int c0 = 0; c0 < Cin; c0 += CONV_CK) {
        const int parts = (Cin - c0 >= CONV_CK) ? 4 : (Cin - c0) / 8;   // 16-byte pieces per row in this chunk
        const int ksteps = (parts + 1) / 2;
        __syncthreads();   // previous chunk fully consumed (and s_pl visible on the first trip)
        if (c0 == 0) CONV_STAMP(1);
        // -- stage the patch AND the weight rows of this chunk in one batched round: every thread first issues all of
        //    its global loads (up to STAGE_U x 16 B), then applies the fused input transforms and writes LDS.
        const bool src1 = c0 >= a.C0;
        const bf16* src = src1 ? a.in1 : a.in0;
        const int cs = src1 ? a.C1 : a.C0, cc = src1 ? c0 - a.C0 : c0;
        const bool xform = !src1 && (a.in_add || a.in_mask || (PRE && a.in_gn));
        // one item = one 16-byte piece (part) of a row; rows 0 .. npix-1 are patch pixels, the rest weight rows.  A
        // thread keeps its piece index and walks rows in steps of 64, all in 32-bit arithmetic (the host checks the ranges).
        const int part = tid & 3, n_rows = npix + ntaps * BN;
        const int w_lds = (int)(s_w - s_patch) - npix * CONV_RS;
        if (part < parts) {
            for (int rb = tid >> 2; rb < n_rows; rb += 64 * STAGE_U) {
                bf16x8 v[STAGE_U];
                int eo[STAGE_U], pl[STAGE_U], ldst[STAGE_U];
#pragma unroll
                for (int u = 0; u < STAGE_U; ++u) {
                    const int row = rb + u * 64;
                    pl[u] = -1; ldst[u] = -1; eo[u] = 0;
                    if (row < npix) {
                        pl[u] = s_pl[row];
                        ldst[u] = row * CONV_RS + part * 16;
                        eo[u] = pl[u] * cs + cc + part * 8;
                    } else if (row < n_rows) {
                        const int wr = row - npix;
                        const int j = wr / BN, co = wr - j * BN;
                        pl[u] = -2;          // weight row
                        ldst[u] = w_lds + row * CONV_RS + part * 16;
                        eo[u] = (s_tap[j] * a.w_rows + co0 + co) * Cin + c0 + part * 8;
                    }
                }
#pragma unroll
                for (int u = 0; u < STAGE_U; ++u) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
                    if (pl[u] >= 0) v[u] = *(const bf16x8*)(src + eo[u]);
                    else if (pl[u] == -2) v[u] = *(const bf16x8*)(a.w + eo[u]);
                }
                if (xform) {
#pragma unroll
                    for (int u = 0; u < STAGE_U; ++u) {
                        if (pl[u] < 0) continue;
                        if constexpr (PRE) {
                            if (a.in_gn) {
                                // normalise + activate what the producer left raw, rounded to bf16 as a GroupNorm launch would
                                // have stored it (padding pixels stay zero: the padding belongs to the normalised tensor)
                                const int n = n0 + div_fast(rb + u * 64, g.phw_magic);
                                const f32x4* af = (const f32x4*)(a.in_gn + ((long long)n * a.C0 + cc + part * 8) * 2);
                                float sc[8], sh[8];
#pragma unroll
                                for (int k = 0; k < 4; ++k) { const f32x4 t = af[k]; sc[2 * k] = t[0]; sh[2 * k] = t[1]; sc[2 * k + 1] = t[2]; sh[2 * k + 1] = t[3]; }
                                if (a.in_gn_act == ACT_GELU) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[u][j] = (bf16)act_fwd((float)v[u][j] * sc[j] + sh[j], ACT_GELU);
                                } else if (a.in_gn_act == ACT_SILU) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[u][j] = (bf16)act_fwd((float)v[u][j] * sc[j] + sh[j], ACT_SILU);
                                } else {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) v[u][j] = (bf16)((float)v[u][j] * sc[j] + sh[j]);
                                }
                            }
                        }
                        if (a.in_add) {
                            const int n = n0 + div_fast(rb + u * 64, g.phw_magic);
                            const float* ad = a.in_add + n * a.C0 + cc + part * 8;
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[u][j] = (bf16)((float)v[u][j] + ad[j]);
                        }
                        if (a.in_mask) {
                            const unsigned m = a.in_mask[eo[u] >> 3];
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[u][j] = (m >> j) & 1 ? (bf16)((float)v[u][j] * a.in_mask_scale) : (bf16)0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < STAGE_U; ++u)
                    if (ldst[u] >= 0) *(bf16x8*)(s_patch + ldst[u]) = v[u];
            }
        }
        __syncthreads();
        if (c0 == 0) CONV_STAMP(2);
        // -- MFMA over taps x k-steps, software pipelined: the operand fragments of step s + D are requested right after
        //    the MFMAs of step s have been issued, so the LDS latency of D steps overlaps the matrix pipeline
        {
            constexpr int D = NT == 1 ? 4 : 2;
            const int total = ntaps * ksteps, kshift = ksteps - 1;         // ksteps is 1 or 2
            bf16x8 fa[D], fb[D][NT];
            auto fetch = [&](int st, bf16x8& av, bf16x8 (&bv)[NT]) {
                const int j = st >> kshift, kk = st & kshift;
                const int poff = __builtin_amdgcn_readlane(my_poff, j);
                av = *(const bf16x8*)(s_patch + a_base + poff * CONV_RS + kk * 32);
                const char* bp = s_w + j * BN * CONV_RS + b_base + kk * 32;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = *(const bf16x8*)(bp + nt * 32 * CONV_RS);
            };
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (d < total) fetch(d, fa[d], fb[d]);
            for (int s0 = 0; s0 < total; s0 += D) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int st = s0 + d;
                    if (st < total) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[d], fb[d][nt], acc[nt], 0, 0, 0);
                        if (st + D < total) fetch(st + D, fa[d], fb[d]);
                    }
                }
            }
        }
        if (c0 == 0) CONV_STAMP(3);
    }

    CONV_STAMP(4);
    // ---- epilogue
    const int offh = g.cls_off_h[cls], offw = g.cls_off_w[cls];
    if (a.sum_out) {
        // per-image spatial sums from the fp32 accumulators (before any rounding, mask or residual)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = co0 + wave_n * BNW + nt * 32 + (lane & 31);
            const float b = bias_r[nt];
            float run = 0.f;
            int run_n = -1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ril = row >> g.thw_shift, rr = row - (ril << g.thw_shift);
                const int rty = rr >> g.tw_shift, rtx = rr - (rty << g.tw_shift);
                const int n = n0 + ril, py = oy + rty, px = ox + rtx;
                const bool ok = n < g.N && py < g.GH && px < g.GW && py * g.SO + offh < g.OH && px * g.SO + offw < g.OW;
                if (!ok) continue;
                if (n != run_n) {
                    if (run_n >= 0 && co < a.Cout) atomicAdd(a.sum_out + (long long)run_n * a.Cout + co, run);
                    run = 0.f; run_n = n;
                }
                run += act_fwd(acc[nt][r] + b, a.act);
            }
            if (run_n >= 0 && co < a.Cout) atomicAdd(a.sum_out + (long long)run_n * a.Cout + co, run);
        }
    }
    if (a.out_mode == OUT_F32_NCHW) {
        // few output channels (the image head): scattered fp32 stores straight from the accumulators
        float* out = (float*)a.out;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = co0 + wave_n * BNW + nt * 32 + (lane & 31);
            if (co >= a.Cout) continue;
            const float b = bias_r[nt];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ril = row >> g.thw_shift, rr = row - (ril << g.thw_shift);
                const int rty = rr >> g.tw_shift, rtx = rr - (rty << g.tw_shift);
                const int n = n0 + ril, py = oy + rty, px = ox + rtx;
                const int oh = py * g.SO + offh, ow = px * g.SO + offw;
                if (n < g.N && py < g.GH && px < g.GW && oh < g.OH && ow < g.OW)
                    out[(((long long)n * a.Cout + co) * g.OH + oh) * g.OW + ow] = act_fwd(acc[nt][r] + b, a.act);
            }
        }
        return;
    }
    CONV_STAMP(5);
    __syncthreads();   // every wave is done reading the operand tiles; reuse LDS for the output tile
    char* s_out = smem;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = wave_n * BNW + nt * 32 + (lane & 31);
        const float b = bias_r[nt];
        // the activation code is uniform: decide once per tile, not inside the unrolled element loop (a switch there is a
        // scalar branch per element)
        if (a.act == ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_out + row * OS + col * 2) = (bf16)(acc[nt][r] + b);
            }
        } else if (a.act == ACT_GELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_out + row * OS + col * 2) = (bf16)act_fwd(acc[nt][r] + b, ACT_GELU);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_out + row * OS + col * 2) = (bf16)act_fwd(acc[nt][r] + b, a.act);
            }
        }
    }
    __syncthreads();
    CONV_STAMP(6);
    bf16* out = (bf16*)a.out;
    float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_ss[4] = {0.f, 0.f, 0.f, 0.f};   // STATS: sums of this thread's four channel pairs (its
                                                                            // 8-channel column is the same for all its rows)
    for (int it = tid; it < BM * (BN / 8); it += T) {
        const int row = it / (BN / 8), part = it - row * (BN / 8);
        const int co = co0 + part * 8;
        if (co >= a.Cout) continue;
        const int ril = row >> g.thw_shift, rr = row - (ril << g.thw_shift);
        const int rty = rr >> g.tw_shift, rtx = rr - (rty << g.tw_shift);
        const int n = n0 + ril, py = oy + rty, px = ox + rtx;
        const int oh = py * g.SO + offh, ow = px * g.SO + offw;
        if (!(n < g.N && py < g.GH && px < g.GW && oh < g.OH && ow < g.OW)) continue;
        const long long e = ((long long)(n * g.OH + oh) * g.OW + ow) * a.out_cs + co;
        bf16x8 v = *(const bf16x8*)(s_out + row * OS + part * 16);
        if constexpr (STATS) {
            pair_sums(v, st_s, st_ss);
        }
        if (a.out_mask) {
            const unsigned m = a.out_mask[e >> 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (m >> j) & 1 ? (bf16)((float)v[j] * a.out_mask_scale) : (bf16)0.f;
        }
        if (a.res) {
            const bf16x8 rv = *(const bf16x8*)(a.res + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)rv[j]);
        }
        if (co + 8 <= a.Cout) {
            *(bf16x8*)(out + e) = v;
        } else {
            for (int j = 0; j < a.Cout - co; ++j) out[e + j] = v[j];
        }
        if (a.out2) {                  // Cout % 8 == 0 (host-checked)
            bf16x8 y;
            if (a.act2 == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)v[j], ACT_GELU);
            } else if (a.act2 == ACT_SILU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)v[j], ACT_SILU);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = (bf16)act_fwd((float)v[j], a.act2);
            }
            *(bf16x8*)(a.out2 + e) = y;
        }
    }
    if constexpr (STATS) {
        // Every tile lies in one image and holds whole groups of 2, 4, 8, ... channels (host-checked).  Lanes that share a
        // column add up by shuffles, the four waves' column sums meet in LDS, and one thread per group adds its channel pairs
        // in a fixed order (deterministic) and stores the tile's pair.
        constexpr int PARTS = BN / 8;
#pragma unroll
        for (int o = PARTS; o < 64; o <<= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { st_s[q] += __shfl_xor(st_s[q], o); st_ss[q] += __shfl_xor(st_ss[q], o); }
        }
        __syncthreads();                               // the output tile in LDS has been read by everyone
        float* red = (float*)smem;                     // [wave][column][4 sums, 4 sums of squares]
        if (lane < PARTS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { red[((tid >> 6) * PARTS + lane) * 8 + q] = st_s[q]; red[((tid >> 6) * PARTS + lane) * 8 + 4 + q] = st_ss[q]; }
        }
        __syncthreads();
        const int cpg = a.Cout / a.gn_G;
        if (tid < BN / cpg) {
            float S = 0.f, SS = 0.f;
            for (int c2 = tid * cpg / 2; c2 < (tid + 1) * cpg / 2; ++c2) {
                const int col = c2 >> 2, q = c2 & 3;
#pragma unroll
                for (int w = 0; w < T / 64; ++w) { S += red[(w * PARTS + col) * 8 + q]; SS += red[(w * PARTS + col) * 8 + 4 + q]; }
            }
            const int tile_in_image = (cls * g.tiles_h + thi) * g.tiles_w + twi;
            float* o = a.gn_part + (((long long)n0 * a.gn_G + co0 / cpg + tid) * a.gn_PT + tile_in_image) * 2;
            o[0] = S; o[1] = SS;
        }
    }
    CONV_STAMP(7);
}

// ------------------------------------------------------------------------------------------------ host
template <int WM, int WN, int NT, bool STATS = false, bool PRE = false>
static int launch_conv(const ConvArgs& a, size_t lds, hipStream_t stream) {
    const ConvGeom& g = a.g;
    auto k = conv_mfma_kernel<WM, WN, NT, STATS, PRE>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ddpm_check_launch("conv: raise dynamic LDS limit");
    }
    dim3 grid(g.tiles_n * g.tiles_h * g.tiles_w, a.CoutW / (32 * NT * WN), g.ncls);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream, a);
    return ddpm_check_launch("conv_mfma");
}

// Test hook: the tile the next general-kernel launches of this thread must use (0 = choose by shape) and the tile the last
// one used, both coded wm * 100 + wn * 10 + nt.
static thread_local int g_force_tile = 0, g_last_tile = 0, g_last_kernel = 0;
extern "C" int ddpm_conv_last_kernel(void) { return g_last_kernel; }
extern "C" int ddpm_conv_debug_tile(int force) {
    const int last = g_last_tile;
    if (force >= 0) g_force_tile = force;
    return last;
}

static int conv_general(const ddpm_conv_desc* d, hipStream_t stream);
extern "C" int ddpm_conv_gn_part_tiles(const ddpm_conv_desc* d);

extern "C" int ddpm_conv_forward(const ddpm_conv_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(d && d->in0 && d->w && (d->out || (d->gn_gamma && d->gn_out && !d->out2 && !d->gn_part)),
                 "conv: null pointer (the raw output may be omitted only with a fused GroupNorm, whose gn_out is the result)");
    DDPM_REQUIRE(d->kind >= 0 && d->kind < CONV_KIND_COUNT, "conv: unknown kind %d", d->kind);
    DDPM_REQUIRE(d->N > 0 && d->IH > 0 && d->IW > 0, "conv: bad input dims %d x %d x %d", d->N, d->IH, d->IW);
    DDPM_REQUIRE(d->C0 > 0 && d->C0 % 16 == 0 && d->C1 >= 0 && d->C1 % 32 == 0 && (d->C1 == 0 || (d->in1 && d->C0 % 32 == 0)),
                 "conv: input channels must be multiples of 16 (32 when two sources are concatenated); got %d + %d", d->C0, d->C1);
    DDPM_REQUIRE(d->Cout > 0 && d->CoutW >= d->Cout && d->CoutW % 32 == 0, "conv: CoutW (%d) must be a multiple of 32 >= Cout (%d)",
                 d->CoutW, d->Cout);
    DDPM_REQUIRE(d->w_rows >= d->CoutW, "conv: w_rows (%d) < CoutW (%d)", d->w_rows, d->CoutW);
    DDPM_REQUIRE(d->out_mode == OUT_F32_NCHW || (d->out_cs >= d->Cout && d->out_cs % 8 == 0),
                 "conv: NHWC output channel stride %d must be a multiple of 8 >= Cout", d->out_cs);
    DDPM_REQUIRE(!(d->kind == DGRAD_CONVT_2X2) || (d->IH % 2 == 0 && d->IW % 2 == 0), "conv: convT data-gradient needs even dims");
    DDPM_REQUIRE((long long)d->N * d->IH * d->IW * (d->C0 > d->C1 ? d->C0 : d->C1) < (1ll << 31), "conv: tensor too large for 32-bit pixel index");
    DDPM_REQUIRE(!(d->gn_gamma && ACT_HAS_BETA(d->gn_act)) || d->gn_act_beta, "conv: a fused GroupNorm with swish / hardswish needs the activation's scalar (gn_act_beta)");
    // pointwise convolutions of large images: the streaming kernel (weights in registers, no workgroup barrier), unless a test
    // has forced a tile of the general kernel
    if (const int rc = g_force_tile ? 0 : ddpm_conv_pw_stream_try(d, stream)) { g_last_kernel = DDPM_CONV_KERNEL_PW_STREAM; return rc > 0 ? DDPM_OK : DDPM_EHIP; }
    if (d->in_affine && !d->gn_part) {
        // otherwise normalise-on-load exists in the general kernel only
        DDPM_REQUIRE(!d->gn_gamma && !d->gnb_raw, "conv: normalise-on-load does not combine with the fused GroupNorm launches");
        return conv_general(d, stream);
    }
    if (d->gn_part) {
        // per-tile GroupNorm sums come from the general kernel's epilogue only (large images, where the GroupNorm that follows
        // takes its many-workgroup path): the caller has asked ddpm_conv_gn_part_tiles first
        DDPM_REQUIRE(ddpm_conv_gn_part_tiles(d) > 0, "conv: no per-tile GroupNorm sums for this problem (ask ddpm_conv_gn_part_tiles first)");
        return conv_general(d, stream);
    }
    // deep, wide convolutions at batch sizes that fill the machine: the pipelined 256 x 128 tile kernel (matrix-core bound)
    if (const int rc = ddpm_conv_pipe_try(d, stream)) { g_last_kernel = DDPM_CONV_KERNEL_PIPE; return rc > 0 ? DDPM_OK : DDPM_EHIP; }
    // 3x3 convolutions on 2x2 / 4x4 images (the deepest stages) have their own kernel
    if (const int rc = ddpm_conv_small_try(d, stream)) { g_last_kernel = DDPM_CONV_KERNEL_SMALL; return rc > 0 ? DDPM_OK : DDPM_EHIP; }
    // whole-image tiles: the unit's GroupNorm (forward) or its backward (data-gradient launches) runs in the same launch
    if (const int rc = ddpm_conv_img_try(d, stream, 0)) { g_last_kernel = DDPM_CONV_KERNEL_IMAGE; return rc > 0 ? DDPM_OK : DDPM_EHIP; }
    DDPM_REQUIRE(!d->gnb_raw, "conv: no fused GroupNorm backward for this shape (ask ddpm_conv_gnb_fusable first)");
    // the image head (a handful of output channels) runs on the vector ALUs
    if (const int rc = ddpm_conv_thin_try(d, stream)) { g_last_kernel = DDPM_CONV_KERNEL_THIN; return rc > 0 ? DDPM_OK : DDPM_EHIP; }
    DDPM_REQUIRE(!d->gn_gamma, "conv: no fused GroupNorm for this shape (ask ddpm_conv_gn_fusable first)");
    return conv_general(d, stream);
}

struct ConvCfg { int wm, wn, nt; };

// Tile of the general kernel for a problem (and its geometry in *g): -1 when none fits.
static int conv_pick(const ddpm_conv_desc* d, ConvCfg* out, ConvGeom* g) {
    const int kind_taps = (d->kind == CONV_3X3_S1 || d->kind == CONV_3X3_S2) ? 9 : 4;
    const int co32 = d->CoutW / 32;
    const ConvCfg cands[] = {{4, 1, 4}, {4, 1, 2}, {2, 2, 2}, {4, 1, 1}, {2, 2, 1}};
    int pick = -1;
    for (int i = 0; i < 5; ++i) {
        const ConvCfg c = cands[i];
        const int bn32 = c.nt * c.wn;
        if (co32 % bn32) continue;
        if (kind_taps >= 9 && bn32 > 2) continue;                 // 9 taps x 128 rows of weights would not fit LDS comfortably
        if (conv_geom_init(g, d->kind, d->N, d->IH, d->IW, 32 * c.wm) != 0) return -2;
        const long long blocks = (long long)g->tiles_n * g->tiles_h * g->tiles_w * (co32 / bn32) * g->ncls;
        pick = i;
        if (g_force_tile ? g_force_tile == c.wm * 100 + c.wn * 10 + c.nt : blocks >= 384) break;
        if (g_force_tile) pick = -1;
    }
    if (pick < 0) return -1;
    *out = cands[pick];
    conv_geom_init(g, d->kind, d->N, d->IH, d->IW, 32 * out->wm);
    return pick;
}

// Partial-sum slots per (image, group) that a launch with gn_part writes (0: this problem cannot deliver them): the general
// kernel's tiles must each lie inside one image, cover only valid positions and hold whole groups of 2, 4, 8, ... channels.
extern "C" int ddpm_conv_gn_part_tiles(const ddpm_conv_desc* d) {
    if (!d || d->gn_part_groups <= 0) return 0;
    if (d->kind != CONV_3X3_S1 && d->kind != CONV_1X1 && d->kind != CONVT_2X2_S2) return 0;
    if (d->out_mode != OUT_BF16_NHWC || d->out_mask || d->res || d->act != 0 || d->Cout != d->CoutW || d->Cout % d->gn_part_groups) return 0;
    if (d->gn_gamma || d->gnb_raw || d->out2 || d->sum_out || d->C0_used) return 0;
    if (d->N <= 0 || d->IH <= 0 || d->IW <= 0 || d->CoutW <= 0 || d->CoutW % 32) return 0;
    const int cpg = d->Cout / d->gn_part_groups;
    if (cpg < 2 || (cpg & (cpg - 1))) return 0;
    // only where the GroupNorm that follows takes its many-workgroup path (more than 4096 vectors per image and channel slab,
    // the rule of groupnorm.hip); smaller problems run the single-pass GroupNorm, which computes its own statistics
    const int slab = d->Cout % 64 == 0 ? 64 : (d->Cout % 32 == 0 ? 32 : (d->Cout % 16 == 0 ? 16 : 8));
    if (d->kind == CONV_1X1 && (long long)d->IH * d->IW * (slab / 8) > 4096 && d->IH > 8 && d->IW > 8) {
        if (const int t = g_force_tile ? 0 : ddpm_conv_pw_stream_tiles(d)) return t;      // the streaming kernel: one slot per workgroup
    }
    ConvCfg cfg; ConvGeom g;
    if (conv_pick(d, &cfg, &g) < 0) return 0;
    const int bn = 32 * cfg.nt * cfg.wn;
    if (bn % cpg || g.NB != 1 || g.GH % g.TH || g.GW % g.TW) return 0;
    if ((long long)g.OH * g.OW * (slab / 8) <= 4096 || g.OH <= 8 || g.OW <= 8) return 0;
    return g.ncls * g.tiles_h * g.tiles_w;
}

static int conv_general(const ddpm_conv_desc* d, hipStream_t stream) {
    ConvArgs a;
    // Tile choice.  Workgroup = 4 waves as WM x WN; tile = (32*WM) positions x (32*NT*WN) output channels.
    // Start from the largest tile (most operand reuse) and shrink it until the launch has >= 384 workgroups
    // (1.5 per CU) or cannot shrink further: these layers are latency/bandwidth bound, so parallelism wins.
    ConvCfg cfg;
    const int pick = conv_pick(d, &cfg, &a.g);
    DDPM_REQUIRE(pick != -2, "conv: geometry");
    DDPM_REQUIRE(pick >= 0 || !g_force_tile, "conv: forced tile %d is not available for CoutW=%d, kind %d", g_force_tile, d->CoutW, d->kind);
    DDPM_REQUIRE(pick >= 0, "conv: no tile configuration for CoutW=%d", d->CoutW);
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.C0 = d->C0; a.C1 = d->C1;
    a.in_add = d->in_add; a.in_mask = d->in_mask; a.in_mask_scale = d->in_mask_scale;
    a.w = (const bf16*)d->w; a.bias = d->bias; a.out = d->out; a.res = (const bf16*)d->res;
    a.out_mask = d->out_mask; a.out_mask_scale = d->out_mask_scale;
    a.Cout = d->Cout; a.CoutW = d->CoutW; a.out_cs = d->out_cs; a.w_rows = d->w_rows; a.act = d->act; a.out_mode = d->out_mode; a.sum_out = d->sum_out;
    a.out2 = (bf16*)d->out2; a.act2 = d->act2;
    a.in_gn = d->in_affine; a.in_gn_act = d->in_affine_act;
    DDPM_REQUIRE(!d->in_affine || (!d->in_add && d->C1 == 0 && (d->in_affine_act == ACT_NONE || d->in_affine_act == ACT_GELU || d->in_affine_act == ACT_SILU)),
                 "conv: normalise-on-load takes one input tensor, no broadcast add, and none / gelu / silu as activation");
    a.gn_part = d->gn_part; a.gn_G = d->gn_part_groups; a.gn_PT = a.g.ncls * a.g.tiles_h * a.g.tiles_w;
    DDPM_REQUIRE(!d->out2 || (d->out_mode == OUT_BF16_NHWC && d->Cout % 8 == 0 && d->act == 0), "conv: the second (activated) result needs an NHWC bf16 output, Cout %% 8 == 0 and no first activation");
    DDPM_REQUIRE(!ACT_HAS_BETA(d->act) && !ACT_HAS_BETA(d->act2) && !ACT_HAS_BETA(d->gn_act) && !ACT_HAS_BETA(d->gnb_act),
                 "conv: swish / hardswish (learnable scalar) are not fused into the convolution launches; apply them with ddpm_gn_act_forward / _backward");

    const int npix = a.g.NB * a.g.PH * a.g.PW;
    const int bm = 32 * cfg.wm, bn = 32 * cfg.nt * cfg.wn;
    g_last_tile = cfg.wm * 100 + cfg.wn * 10 + cfg.nt;
    g_last_kernel = DDPM_CONV_KERNEL_GENERAL;
    size_t lds = ((size_t)npix * 4 + 15) / 16 * 16 + CONV_MAX_TAPS * 4 + (size_t)npix * CONV_RS + (size_t)a.g.max_taps * bn * CONV_RS;
    const size_t lds_out = (size_t)bm * (bn * 2 + 16);
    if (lds < lds_out) lds = lds_out;
    DDPM_REQUIRE(lds <= 160 * 1024, "conv: tile needs %zu B of LDS", lds);

    if (a.in_gn) {
        if (a.gn_part && lds < 4 * 16 * 8 * sizeof(float)) lds = 4 * 16 * 8 * sizeof(float);
#define CONV_CASE(WMv, WNv, NTv) if (cfg.wm == WMv && cfg.wn == WNv && cfg.nt == NTv) return a.gn_part ? launch_conv<WMv, WNv, NTv, true, true>(a, lds, stream) : launch_conv<WMv, WNv, NTv, false, true>(a, lds, stream);
        CONV_CASE(4, 1, 4) CONV_CASE(4, 1, 2) CONV_CASE(2, 2, 2) CONV_CASE(4, 1, 1) CONV_CASE(2, 2, 1)
#undef CONV_CASE
    }
    if (a.gn_part) {
        if (lds < 4 * 16 * 8 * sizeof(float)) lds = 4 * 16 * 8 * sizeof(float);
#define CONV_CASE(WMv, WNv, NTv) if (cfg.wm == WMv && cfg.wn == WNv && cfg.nt == NTv) return launch_conv<WMv, WNv, NTv, true>(a, lds, stream);
        CONV_CASE(4, 1, 4) CONV_CASE(4, 1, 2) CONV_CASE(2, 2, 2) CONV_CASE(4, 1, 1) CONV_CASE(2, 2, 1)
#undef CONV_CASE
    }
#define CONV_CASE(WMv, WNv, NTv) if (cfg.wm == WMv && cfg.wn == WNv && cfg.nt == NTv) return launch_conv<WMv, WNv, NTv>(a, lds, stream);
    CONV_CASE(4, 1, 4) CONV_CASE(4, 1, 2) CONV_CASE(2, 2, 2) CONV_CASE(4, 1, 1) CONV_CASE(2, 2, 1)
#undef CONV_CASE
    ddpm_set_error("conv: no kernel for tile %dx%dx%d", cfg.wm, cfg.wn, cfg.nt);
    return DDPM_EINVAL;
}
