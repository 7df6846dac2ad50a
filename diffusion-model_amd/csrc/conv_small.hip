// 3x3 stride-1 convolution (forward or data-gradient form) on images of 2x2, 4x4 or 8x8 pixels.
//
// The deepest UNet stages run 3x3 convolutions with 128-512 channels on a few pixels per image: a GEMM with only
// N*H*W = 1-4 K rows but K = 9*Cin up to 4608 deep.  The general tap-list kernel walks K in 32-channel chunks with a
// stage/barrier/MFMA round per chunk and has too few workgroups to hide that; here a workgroup owns a 32-row x 32-column
// output tile, keeps the input rows of its images in LDS once (plus one zero row that all out-of-image taps read), and
// its four waves split the input channels four ways, each streaming its weight fragments straight from L2 with a tap
// of look-ahead; the four partial tiles are added through LDS.  Same operands, fused input transforms (channel concat,
// time-embedding add) and epilogue (bias, activation, residual, per-image sums) as the general kernel.
//
// A 32-row tile is 8 (2x2) or 2 (4x4) whole images and 32 output channels are whole normalisation groups, so the
// GroupNorm + activation that follows every such convolution in the UNet (conv_block.py:19-54 there) can finish in
// the same launch: `out` still receives the raw result (the backward pass needs it), gn_out the normalised one.
#include "conv_geom.h"
#include "ddpm_hip.h"

struct SmallConvArgs {
    const bf16* in0; const bf16* in1; int C0, C1;
    const float* in_add;
    const bf16* w; const float* bias;
    bf16* out; const bf16* res;
    float* sum_out;
    int N, Cout, out_cs, w_rows, act;
    // optional fused GroupNorm + activation (+ residual) of the result: a tile holds whole images and whole groups
    const float* gn_gamma; const float* gn_beta; bf16* gn_out; float* gn_stats; const bf16* gn_res;
    int gn_groups, gn_act; float gn_eps;
    const float* gn_act_beta;        // swish / hardswish: the activation's learnable scalar (else null)
    // optional GroupNorm + activation BACKWARD applied to the input on its way into LDS (data-gradient launches): in0 is then
    // the gradient of the normalised, activated tensor; the tile holds whole images and all channels, so the statistics of
    // the backward pass are complete inside the workgroup
    const bf16* gnb_raw; const float* gnb_gamma; const float* gnb_beta; const float* gnb_stats; float* gnb_part; bf16* gnb_draw;
    int gnb_groups, gnb_act;
};

// KS = MFMA k-steps per wave and tap (Cin = 64 * KS).  The weight fragments of a whole group of taps are requested at once
// (all nine for KS <= 4, three at a time - double buffered - for KS = 8): one L2 round trip per group instead of one per tap.
// A: the activation of the fused GroupNorm (forward epilogue or backward loader) as a compile-time code for the hot ones
// (gelu, silu, none), -1 = the runtime code: a runtime switch inside the unrolled element loops costs a scalar branch per
// element and keeps the compiler from packing the arithmetic.
template <int W, int KS, int A>
__global__ __launch_bounds__(256) void conv_small_kernel(const SmallConvArgs a) {
    constexpr int HW = W * W;
    constexpr int MB = HW > 32 ? HW / 32 : 1, ROWS = 32 * MB;     // 32-row MFMA blocks per tile: an 8x8 image is two
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cin = a.C0 + a.C1;
    const int AS = Cin * 2 + 16;
    char* s_a = smem;                                  // [ROWS + 1][AS]; the last row stays zero
    float* s_red = (float*)(smem + (ROWS + 1) * AS);   // [4][ROWS][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = (long long)blockIdx.x * ROWS;
    const long long rows_total = (long long)a.N * HW;
    const int co0 = blockIdx.y * 32;

    // ---- input rows of this tile's images -> LDS (concat of two sources; broadcast add on source 0)
    const int vpr = Cin / 8;
    if (a.gnb_raw) {
        // GroupNorm + activation backward on the way in.  unit = (image of the tile, 8-channel vector), tpu threads per
        // unit share its HW rows; with g = dy act'(z), xh = (x - mean) rstd:  dgamma = sum g xh, dbeta = sum g,
        // dx = rstd (gamma g - (S1 + xh S2) / m),  S1 = sum_group gamma g, S2 = sum_group gamma g xh,  m = cpg HW.
        constexpr int IPT = ROWS / HW;
        const int C = Cin, G = a.gnb_groups, cpg = C / G;
        const int units = IPT * vpr, tpu = 256 / units, rpt = HW / tpu;     // host-checked: 4 <= units <= 256, rpt in {1, 2, 4}
        const int unit = tid / tpu, sub = tid - unit * tpu;
        const int img = unit / vpr, vec = unit - img * vpr;
        const long long n = row0 / HW + img;
        const bool have = n < a.N;
        float ga[8], be[8], mean[8], rstd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = vec * 8 + j, grp = c / cpg;
            ga[j] = a.gnb_gamma[c]; be[j] = a.gnb_beta[c];
            mean[j] = have ? a.gnb_stats[(n * G + grp) * 2] : 0.f;
            rstd[j] = have ? a.gnb_stats[(n * G + grp) * 2 + 1] : 0.f;
        }
        float gv[4][8], xh[4][8];
        float a1[8], a2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < rpt) {
                const long long e = (row0 + img * HW + sub * rpt + k) * C + vec * 8;
                bf16x8 xv, dv;
#pragma unroll
                for (int j = 0; j < 8; ++j) { xv[j] = (bf16)0.f; dv[j] = (bf16)0.f; }
                if (have) { xv = *(const bf16x8*)(a.gnb_raw + e); dv = *(const bf16x8*)(a.in0 + e); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[k][j] = ((float)xv[j] - mean[j]) * rstd[j];
                    gv[k][j] = (float)dv[j] * act_bwd(xh[k][j] * ga[j] + be[j], A >= 0 ? A : a.gnb_act);
                    a1[j] += gv[k][j]; a2[j] += gv[k][j] * xh[k][j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = 1; o < tpu; o <<= 1) { a1[j] += __shfl_xor(a1[j], o, 64); a2[j] += __shfl_xor(a2[j], o, 64); }
        if (have && sub == 0 && blockIdx.y == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                a.gnb_part[(n * 3 + 0) * C + vec * 8 + j] = a2[j];
                a.gnb_part[(n * 3 + 1) * C + vec * 8 + j] = a1[j];
            }
        }
        float S1[8], S2[8];
        const float im = 1.0f / (float)(cpg * HW);
        if (cpg >= 8) {
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { u += a1[j] * ga[j]; v += a2[j] * ga[j]; }
            for (int o = tpu; o < tpu * (cpg / 8); o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
#pragma unroll
            for (int j = 0; j < 8; ++j) { S1[j] = u * im; S2[j] = v * im; }
        } else {                                       // cpg == 4: two groups inside the vector
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float u = 0.f, v = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { u += a1[h * 4 + j] * ga[h * 4 + j]; v += a2[h * 4 + j] * ga[h * 4 + j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) { S1[h * 4 + j] = u * im; S2[h * 4 + j] = v * im; }
            }
        }
        float db[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) db[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < rpt) {
                const int row = img * HW + sub * rpt + k;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dd = rstd[j] * (ga[j] * gv[k][j] - (S1[j] + xh[k][j] * S2[j]));
                    db[j] += dd;
                    o[j] = (bf16)dd;
                }
                *(bf16x8*)(s_a + row * AS + vec * 16) = o;
                if (have && blockIdx.y == 0) *(bf16x8*)(a.gnb_draw + (row0 + row) * C + vec * 8) = o;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = 1; o < tpu; o <<= 1) db[j] += __shfl_xor(db[j], o, 64);
        if (have && sub == 0 && blockIdx.y == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a.gnb_part[(n * 3 + 2) * C + vec * 8 + j] = db[j];
        }
        for (int idx = tid; idx < vpr; idx += 256) {   // the zero row
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (bf16)0.f;
            *(bf16x8*)(s_a + ROWS * AS + idx * 16) = z;
        }
    } else
    for (int idx = tid; idx < (ROWS + 1) * vpr; idx += 256) {
        const int row = idx / vpr, c = (idx - row * vpr) * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
        if (row < ROWS && row0 + row < rows_total) {
            if (c < a.C0) {
                v = *(const bf16x8*)(a.in0 + (row0 + row) * a.C0 + c);
                if (a.in_add) {
                    const float* ad = a.in_add + ((row0 + row) / HW) * a.C0 + c;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + ad[j]);
                }
            } else {
                v = *(const bf16x8*)(a.in1 + (row0 + row) * a.C1 + (c - a.C0));
            }
        }
        *(bf16x8*)(s_a + row * AS + c * 2) = v;
    }

    // ---- this wave's quarter of the input channels, all nine taps
    constexpr int kq = 16 * KS;                        // channels per wave
    constexpr int TG = KS <= 4 ? 9 : 3, NG = 9 / TG;   // taps per group, groups
    const int r = lane & 31;
    const bf16* wbase = a.w + ((long long)(co0 + r) * Cin + wave * kq + (lane >> 5) * 8);
    const long long slice = (long long)a.w_rows * Cin;
    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
    bf16x8 bw[NG > 1 ? 2 : 1][TG][KS];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int u = 0; u < KS; ++u) bw[0][t][u] = *(const bf16x8*)(wbase + t * slice + u * 16);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int u = 0; u < KS; ++u) bw[(g + 1) & 1][t][u] = *(const bf16x8*)(wbase + ((g + 1) * TG + t) * slice + u * 16);
        }
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int tap = g * TG + t;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int rr = mb * 32 + r;
                const int p = rr % HW, py = p / W, px = p % W;
                const int ny = py + dy, nx = px + dx;
                const int nrow = (ny >= 0 && ny < W && nx >= 0 && nx < W) ? rr + dy * W + dx : ROWS;
                const char* ap = s_a + nrow * AS + (wave * kq + (lane >> 5) * 8) * 2;
#pragma unroll
                for (int u = 0; u < KS; ++u) {
                    const bf16x8 av = *(const bf16x8*)(ap + u * 32);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw[g & 1][t][u], acc[mb], 0, 0, 0);
                }
            }
        }
    }
    // ---- add the four partial tiles
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            s_red[(wave * ROWS + row) * 33 + (lane & 31)] = acc[mb][i];
        }
    __syncthreads();
    const int c4 = (tid & 7) * 4;
    float val[MB][4];
    bool live[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int row = mb * 32 + (tid >> 3);
        const long long grow = row0 + row;
        live[mb] = grow < rows_total;
        const long long e = grow * a.out_cs + co0 + c4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + c4 + j;
            const float v = s_red[(0 * ROWS + row) * 33 + c4 + j] + s_red[(1 * ROWS + row) * 33 + c4 + j] +
                            s_red[(2 * ROWS + row) * 33 + c4 + j] + s_red[(3 * ROWS + row) * 33 + c4 + j];
            val[mb][j] = v + ((a.bias && co < a.Cout) ? a.bias[co] : 0.f);
        }
        if (a.act != ACT_NONE) {             // (a convolution-side activation is the rare case: decided outside the element loop)
#pragma unroll
            for (int j = 0; j < 4; ++j) val[mb][j] = act_fwd(val[mb][j], a.act);
        }
        if (live[mb]) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)val[mb][j];
            if (a.sum_out) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co0 + c4 + j < a.Cout) atomicAdd(a.sum_out + (grow / HW) * a.Cout + co0 + c4 + j, val[mb][j]);
            }
            if (a.res) {
                const bf16x4 rv = *(const bf16x4*)(a.res + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)o[j] + (float)rv[j]);
            }
            if (!a.out) {
                // fused GroupNorm at inference: nobody reads the raw result (host-checked: only with gn_out)
            } else if (co0 + c4 + 4 <= a.Cout) {
                *(bf16x4*)(a.out + e) = o;
            } else {
                for (int j = 0; j < a.Cout - co0 - c4; ++j) a.out[e + j] = o[j];
            }
        }
    }
    if (!a.gn_gamma) return;

    // ---- fused GroupNorm + activation: statistics over the bf16-rounded raw values, as the separate kernel would read them
    const int cpg = a.Cout / a.gn_groups;              // 32 % cpg == 0, Cout % 32 == 0, no conv activation / residual (host-checked)
    const int gpt = 32 / cpg;                          // groups in this tile
    const f32x4 gam = *(const f32x4*)(a.gn_gamma + co0 + c4), bet = *(const f32x4*)(a.gn_beta + co0 + c4);
    bf16x4 rv4[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rv4[mb][j] = (bf16)0.f;
        if (a.gn_res && live[mb]) rv4[mb] = *(const bf16x4*)(a.gn_res + (row0 + mb * 32 + (tid >> 3)) * a.out_cs + co0 + c4);
    }
    __syncthreads();                                   // partial tiles consumed: reuse the table as [ROWS][33] values + [pairs][2] statistics
    float* s_val = s_red;
    float* s_st = s_red + ROWS * 33;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            val[mb][j] = (float)(bf16)val[mb][j];
            s_val[(mb * 32 + (tid >> 3)) * 33 + c4 + j] = live[mb] ? val[mb][j] : 0.f;
        }
    __syncthreads();
    {
        // pair = (image of the tile, group of the tile); eight threads per pair, HW * cpg / 8 values each
        const int pair = tid >> 3, sub = tid & 7;
        const int npairs = (ROWS / HW) * gpt;
        float s1 = 0.f, s2 = 0.f;
        if (pair < npairs) {
            const int img = pair / gpt, grp = pair - img * gpt;
            const int per = HW * cpg / 8;
            for (int q = sub * per; q < (sub + 1) * per; ++q) {
                const int r2 = img * HW + q / cpg, c2 = grp * cpg + q % cpg;
                const float f = s_val[r2 * 33 + c2];
                s1 += f; s2 += f * f;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (pair < npairs && sub == 0) {
            const int img = pair / gpt, grp = pair - img * gpt;
            const float m = 1.0f / (float)(HW * cpg);
            const float mean = s1 * m;
            const float rstd = rsqrtf(fmaxf(s2 * m - mean * mean, 0.f) + a.gn_eps);
            s_st[pair * 2] = mean; s_st[pair * 2 + 1] = rstd;
            const long long n = row0 / HW + img;
            if (n < a.N && a.gn_stats) {
                const int g = co0 / cpg + grp;
                a.gn_stats[(n * a.gn_groups + g) * 2 + 0] = mean;
                a.gn_stats[(n * a.gn_groups + g) * 2 + 1] = rstd;
            }
        }
    }
    __syncthreads();
    const float act_b = a.gn_act_beta ? *a.gn_act_beta : 1.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        if (!live[mb]) continue;
        const int row = mb * 32 + (tid >> 3);
        const int pair = (row / HW) * gpt + c4 / cpg;  // cpg >= 4: the four channels of a thread share a group
        const float mean = s_st[pair * 2], rstd = s_st[pair * 2 + 1];
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sc = gam[j] * rstd, sh = bet[j] - mean * sc;
            float y = act_fwd(val[mb][j] * sc + sh, A >= 0 ? A : a.gn_act, act_b);
            if (a.gn_res) y += (float)rv4[mb][j];
            o[j] = (bf16)y;
        }
        *(bf16x4*)(a.gn_out + (row0 + row) * a.Cout + co0 + c4) = o;
    }
}

int ddpm_conv_img_fusable(const ddpm_conv_desc* d, int backward);     // conv_img.hip

static int small_gnb_fusable(const ddpm_conv_desc* d) {
    if (!d || d->kind != CONV_3X3_S1 || d->IH != d->IW || (d->IW != 2 && d->IW != 4 && d->IW != 8)) return 0;
    if (d->out_mode != OUT_BF16_NHWC || d->in_mask || d->out_mask || d->in_add || d->C1 != 0 || d->out2 || d->gn_gamma) return 0;
    const int C = d->C0, hw = d->IH * d->IW;
    if (C % 64 || C > 512 || (d->IW == 8 && C > 256) || d->out_cs % 4 || d->gnb_groups <= 0 || C % d->gnb_groups) return 0;
    const int cpg = C / d->gnb_groups, rows = hw > 32 ? hw : 32;
    const int units = (rows / hw) * (C / 8);
    if (units < 4 || units > 256 || (256 % units)) return 0;
    const int tpu = 256 / units;
    if (tpu > hw || hw % tpu || hw / tpu > 4) return 0;
    if (cpg == 4) return 1;
    return cpg % 8 == 0 && ((cpg / 8) & (cpg / 8 - 1)) == 0 && tpu * (cpg / 8) <= 64;
}

// Whether the GroupNorm backward described by the gnb_* fields can run in the loader of this (data-gradient) launch.
extern "C" int ddpm_conv_gnb_fusable(const ddpm_conv_desc* d) {
    if (d && ACT_HAS_BETA(d->gnb_act)) return 0;          // an activation with a learnable scalar needs that scalar's gradient too
    return small_gnb_fusable(d) || (d && ddpm_conv_img_fusable(d, 1));
}

static int small_gn_fusable(const ddpm_conv_desc* d) {
    if (!d || d->kind != CONV_3X3_S1 || d->IH != d->IW || (d->IW != 2 && d->IW != 4 && d->IW != 8)) return 0;
    if (d->out_mode != OUT_BF16_NHWC || d->in_mask || d->out_mask || d->res || d->act != 0 || d->sum_out) return 0;
    const int Cin = d->C0 + d->C1;
    if (Cin % 64 || Cin > 512 || d->C0 % 8 || d->C1 % 8 || d->out_cs != d->Cout) return 0;
    if (d->IW == 8 && Cin > 256) return 0;
    if (d->gn_groups <= 0 || d->Cout % 32 || d->Cout % d->gn_groups) return 0;
    const int cpg = d->Cout / d->gn_groups;
    return cpg >= 4 && cpg <= 32 && 32 % cpg == 0;
}

// Whether the GroupNorm described by the gn_* fields can finish inside the convolution's launch.
extern "C" int ddpm_conv_gn_fusable(const ddpm_conv_desc* d) {
    if (d && ACT_HAS_BETA(d->gn_act) && !d->gn_act_beta) return 0;    // swish / hardswish bring their learnable scalar
    return small_gn_fusable(d) || (d && ddpm_conv_img_fusable(d, 0));
}

// Returns 1 when the problem was launched here, 0 when the caller should use the general kernel, < 0 on error.
int ddpm_conv_small_try(const ddpm_conv_desc* d, hipStream_t stream) {
    if (d->kind != CONV_3X3_S1 || d->IH != d->IW || (d->IW != 2 && d->IW != 4 && d->IW != 8)) return 0;
    if (d->out_mode != OUT_BF16_NHWC || d->in_mask || d->out_mask) return 0;
    if (d->out2) return 0;
    const int Cin = d->C0 + d->C1;
    if (Cin % 64 || Cin > 512 || d->C0 % 8 || d->C1 % 8 || d->out_cs % 4) return 0;
    if (d->IW == 8 && Cin > 256) return 0;             // 8x8 images: the UNet's 64-wide stage (two sources when concatenated)
    SmallConvArgs a;
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.C0 = d->C0; a.C1 = d->C1; a.in_add = d->in_add;
    a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (bf16*)d->out; a.res = (const bf16*)d->res; a.sum_out = d->sum_out;
    a.N = d->N; a.Cout = d->Cout; a.out_cs = d->out_cs; a.w_rows = d->w_rows; a.act = d->act;
    a.gn_gamma = d->gn_gamma; a.gn_beta = d->gn_beta; a.gn_out = (bf16*)d->gn_out; a.gn_stats = d->gn_stats; a.gn_res = (const bf16*)d->gn_res;
    a.gn_groups = d->gn_groups; a.gn_act = d->gn_act; a.gn_eps = d->gn_eps; a.gn_act_beta = d->gn_act_beta;
    a.gnb_raw = (const bf16*)d->gnb_raw; a.gnb_gamma = d->gnb_gamma; a.gnb_beta = d->gnb_beta; a.gnb_stats = d->gnb_stats;
    a.gnb_part = d->gnb_part; a.gnb_draw = (bf16*)d->gnb_draw; a.gnb_groups = d->gnb_groups; a.gnb_act = d->gnb_act;
    if (d->gnb_raw && !small_gnb_fusable(d)) return 0;          // another kernel may take it (or the caller gets the error there)
    if (d->gn_gamma && !small_gn_fusable(d)) return 0;
    if (d->gnb_raw && !(d->gnb_gamma && d->gnb_beta && d->gnb_stats && d->gnb_part && d->gnb_draw)) {
        ddpm_set_error("conv: the fused GroupNorm backward needs all gnb_* pointers");
        return -1;
    }
    if (d->gn_gamma && !(d->gn_beta && d->gn_out)) {
        ddpm_set_error("conv: the fused GroupNorm needs gamma, beta and an output");
        return -1;
    }
    const int tile_rows = d->IW == 8 ? 64 : 32;        // whole images: 8 of 2x2, 2 of 4x4, 1 of 8x8
    const size_t lds = (size_t)(tile_rows + 1) * (Cin * 2 + 16) + (size_t)4 * tile_rows * 33 * 4;      // the statistics table of the fused GroupNorm fits the partial-tile area
    const long long rows = (long long)d->N * d->IH * d->IW;
    dim3 grid((unsigned)((rows + tile_rows - 1) / tile_rows), (unsigned)(d->CoutW / 32));
    const int ks = Cin / 64;
    if (ks != 1 && ks != 2 && ks != 4 && ks != 8) return 0;
    const int key = d->gnb_raw ? d->gnb_act : d->gn_gamma ? d->gn_act : ACT_NONE;      // the fused GroupNorm's activation
#define SMALL_LAUNCH(Wv, KSv, Av)                                                                                                \
    {                                                                                                                            \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_small_kernel<Wv, KSv, Av>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess)                                                      \
            return ddpm_check_launch("conv_small: raise dynamic LDS limit") ? -1 : -1;                                           \
        hipLaunchKernelGGL((conv_small_kernel<Wv, KSv, Av>), grid, dim3(256), lds, stream, a);                                    \
    }
#define SMALL_CASE(Wv, KSv)                                                                                                      \
    if (d->IW == Wv && ks == KSv) {                                                                                              \
        if (key == ACT_GELU) SMALL_LAUNCH(Wv, KSv, ACT_GELU)                                                                     \
        else if (key == ACT_SILU) SMALL_LAUNCH(Wv, KSv, ACT_SILU)                                                                \
        else if (key == ACT_NONE) SMALL_LAUNCH(Wv, KSv, ACT_NONE)                                                                \
        else SMALL_LAUNCH(Wv, KSv, -1)                                                                                           \
    }
    SMALL_CASE(2, 1) SMALL_CASE(2, 2) SMALL_CASE(2, 4) SMALL_CASE(2, 8) SMALL_CASE(4, 1) SMALL_CASE(4, 2) SMALL_CASE(4, 4) SMALL_CASE(4, 8)
    SMALL_CASE(8, 1) SMALL_CASE(8, 2) SMALL_CASE(8, 4)
#undef SMALL_CASE
#undef SMALL_LAUNCH
    return ddpm_check_launch("conv_small") == DDPM_OK ? 1 : -1;
}
