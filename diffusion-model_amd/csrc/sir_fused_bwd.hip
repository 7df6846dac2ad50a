// Backward of one SeparableInvertResidual block (conv_block.py:57-136 of the reference) as ONE kernel for the
// data-gradient chain, mirroring sir_fused.hip: a workgroup owns whole images, so the three GroupNorm backward
// reductions are workgroup-local, the two pointwise data-gradients run on MFMA from LDS tiles, and the depthwise
// data-gradient and weight-gradient share one pass over the LDS-resident neighbours:
//
//   dy -> GN3' -> d_o --(x Wp)--> GN2'/act' -> dd --depthwise^T--> GN1'/act' -> d_e --(x We)--> dropout' -> + dy -> dx
//
// Written to memory: dx, d_o and d_e (operands of the two pointwise weight-gradient GEMMs, which sum over the whole
// batch and stay separate launches) and per-image / per-workgroup fp32 partial sums of every per-channel parameter
// gradient (GroupNorm gamma/beta, the three biases, the nine depthwise taps), folded by the caller's reduction pass.
// Everything that feeds the data gradients is summed in a fixed order (per-wave slots), so dx, d_o, d_e and the
// GroupNorm / bias partials are bit-reproducible; only the nine depthwise tap gradients (leaf values) are folded
// across waves with LDS float atomics.
#include "common.h"
#include "ddpm_hip.h"
#include "sir_tile.h"

// Diagnostic build only (-DSIR_STAMPS): lane 0 of every wave records the shader clock at phase boundaries into a buffer of
// its own; no output value depends on it and the product build contains none of this.
#ifdef SIR_STAMPS
__device__ unsigned long long* sir_bwd_dbg = nullptr;
extern "C" int ddpm_sir_bwd_set_dbg(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(sir_bwd_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#define SIRB_STAMP(i)                                                                                      \
    do {                                                                                                   \
        if (sir_bwd_dbg && (threadIdx.x & 63) == 0)                                                        \
            sir_bwd_dbg[((unsigned long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SIRB_STAMP(i)
#endif

struct SirBwdArgs {
    const bf16* dy;
    const bf16* raw_e; const bf16* raw_d; const bf16* raw_o; const bf16* y_e;
    const float* st1; const float* st2; const float* st3;
    const float* g1; const float* b1; const float* g2; const float* b2; const float* g3;
    const bf16* wpd;                 // [E][C]  data-gradient copy of the projection weight (rows = its input channels)
    const bf16* wed;                 // [C][E]  data-gradient copy of the expansion weight
    const float* wdf;                // [9][E]  depthwise taps, flipped
    const unsigned char* m_in; const unsigned char* m_e; const unsigned char* m_d;
    float mscale;
    bf16* dx; bf16* d_o; bf16* d_e;
    float* part1; float* part2; float* part3;      // [N][3][E], [N][3][E], [N][3][C]: d gamma, d beta, column sum of the raw gradient
    float* dwpart;                                 // [workgroups][10][E]: nine tap gradients, then the depthwise bias gradient
    int N, act;
    const float* ab1; const float* ab2;            // swish / hardswish: the two activation modules' learnable scalars ...
    float* dbp1; float* dbp2;                      // ... and [workgroups][waves] partial sums of their gradients
};

template <int ACT>
__device__ __forceinline__ float act_d(float z) {
    if constexpr (ACT == ACT_GELU) { float c, g; gelu_parts(z, c, g); return c + z * 0.39894228040143268f * g; }
    else if constexpr (ACT == ACT_SILU) { const float s = sigmoid_fast(z); return s * (1.0f + z * (1.0f - s)); }
    else if constexpr (ACT == ACT_RELU) return z > 0.f ? 1.f : 0.f;
    else return 1.f;
}

// The same for the codes with a learnable scalar b (swish: z sigmoid(b z), hardswish: z hardsigmoid(b z)), together with the
// derivative with respect to b; the other codes ignore b and return db = 0.
template <int ACT>
__device__ __forceinline__ float act_d_db(float z, float b, float& db) {
    if constexpr (ACT == ACT_SWISH) {
        const float s = sigmoid_fast(b * z), t = s * (1.0f - s);
        db = z * z * t;
        return s + b * z * t;
    } else if constexpr (ACT == ACT_HARDSWISH) {
        // 1/6 inside |u| < 3, else 0, by arithmetic (a huge factor and a clamp to [0, 1]): compares would hold a pair of scalar
        // registers per element until their selects run, more than the scalar file has for a tile of this size
        const float u = b * z;
        const float t = __builtin_amdgcn_fmed3f((3.f - fabsf(u)) * 0x1p60f, 0.f, 1.f) * (1.f / 6.f);
        db = z * z * t;
        return hardsigmoid_of(u) + u * t;
    } else {
        db = 0.f;
        return act_d<ACT>(z);
    }
}

// Sum r[0..7] over the R = 64 / VPR lanes of a wave that hold other rows of the same eight channels.  Every exchange
// halves the set of values a lane keeps (it sends the half its partner keeps), so the whole reduction costs 8 - 8/R
// shuffles instead of 8 * log2(R).  On return lane (g = lane / VPR, vec = lane % VPR) holds in r[0 .. 8/R) the complete
// sums of channels vec * 8 + jb + i, with jb the return value.
template <int VPR>
__device__ __forceinline__ int fold_cols(float (&r)[8], int lane) {
    static_assert(VPR >= 1 && VPR <= 64, "a row of vectors fits a wave");
    // more than eight row groups: plain exchanges over the far ones first (every lane keeps all eight values)
#pragma unroll
    for (int o = 32; o >= VPR * 8; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += __shfl_xor(r[j], o, 64);
    }
    constexpr int TOP = VPR * 8 < 64 ? VPR * 8 : 64;
    int jb = 0;
#pragma unroll
    for (int o = VPR; o < TOP; o <<= 1) {
        const int cnt = 4 * VPR / o;
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < cnt; ++i) {
            const float lo = r[i], hi = r[i + cnt];
            r[i] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, o, 64);
        }
        if (up) jb += cnt;
    }
    return jb;
}

// Deterministic column sums: a thread owns NV consecutive rows, so all rows of a wave belong to one image; the wave
// reduces over its rows with shuffles and stores one value per channel into its own slot [wave][CS]; the finalising
// thread of (image, channel) adds the slots of the image's waves in a fixed order.
template <int CH, int M, int CS>
__device__ __forceinline__ void col_store(const float (&v)[8], int wave, int vec, int lane, float* s_slot) {
    using T = Ew<CH, M>;
    constexpr int KEEP = T::VPR >= 8 ? 8 / (64 / T::VPR) : 1;        // with more than eight row groups lanes hold duplicates: same value, same address
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
    const int jb = fold_cols<T::VPR>(r, lane);
#pragma unroll
    for (int i = 0; i < KEEP; ++i) s_slot[wave * CS + vec * 8 + jb + i] = r[i];
}

template <int IPW, int CS>
__device__ __forceinline__ float col_total(const float* s_slot, int img, int c) {
    constexpr int NWV = SIR_THREADS / 64, WPI = NWV / IPW;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WPI; ++w) t += s_slot[(img * WPI + w) * CS + c];
    return t;
}

// GroupNorm (+activation, +dropout) backward over the vectors a thread holds (rows r0*NV .. r0*NV+NV-1).
// g[][]: in = gradient w.r.t. the (masked, activated) output of this norm, out = gradient w.r.t. the raw convolution
// output.  s_slot: [3][waves][CS] scratch; on return table 2 holds the per-wave column sums of the result (the bias
// gradient of the producing convolution), to be read with col_total after the caller's next barrier.
template <int CH, int M, int HW, int IPW, int CS, int ACT>
__device__ __forceinline__ void norm_bwd(const bf16x8 (&raw)[Ew<CH, M>::NV], float (&g)[Ew<CH, M>::NV][8], const unsigned (&mk)[Ew<CH, M>::NV],
                                         bool use_mask, float mscale, const float* gamma, const float* beta, const float* s_mr,
                                         float* s_slot, float* s_grp, float* part, long long n0, int N, int tid,
                                         float bsc = 1.f, float* dbp = nullptr) {
    // bsc, dbp (swish / hardswish only): the activation's scalar, and this workgroup's [waves] slots for the partial sums of
    // the scalar's gradient = sum of (incoming gradient) * (mask) * d act / d scalar
    using T = Ew<CH, M>;
    constexpr int NWV = SIR_THREADS / 64;
    static_assert(NWV % IPW == 0 && (M / NWV) <= HW, "a wave must not straddle images");
    const int lane = tid & 63, wave = tid >> 6;
    const int vec = tid % T::VPR, r0 = tid / T::VPR;
    const int img = IPW > 1 ? (r0 * T::NV) / HW : 0;          // the same for every row of this thread (and of its wave)
    float ga[8], be[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int grp = (vec * 8 + j) / T::CPG;
        ga[j] = gamma[vec * 8 + j]; be[j] = ACT != ACT_NONE ? beta[vec * 8 + j] : 0.f;
        mean[j] = s_mr[(img * SIR_G + grp) * 2]; rstd[j] = s_mr[(img * SIR_G + grp) * 2 + 1];
    }
    float a1[8], a2[8];
    float dbs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll
    for (int k = 0; k < T::NV; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)raw[k][j] - mean[j]) * rstd[j];
            float gg = g[k][j];
            if constexpr (ACT_HAS_BETA(ACT)) {
                float db;
                const float d = act_d_db<ACT>(xh * ga[j] + be[j], bsc, db);
                db *= gg; gg *= d;
                if (use_mask) db = (mk[k] >> j) & 1 ? db * mscale : 0.f;
                dbs += db;
            } else if constexpr (ACT != ACT_NONE) {
                gg *= act_d<ACT>(xh * ga[j] + be[j]);
            }
            if (use_mask) gg = (mk[k] >> j) & 1 ? gg * mscale : 0.f;
            g[k][j] = gg;
            a1[j] += gg; a2[j] += gg * xh;
        }
    }
    if constexpr (ACT_HAS_BETA(ACT)) {
        dbs = wave_sum(dbs);
        if (lane == 0) dbp[wave] = dbs;
    }
    col_store<CH, M, CS>(a1, wave, vec, lane, s_slot);
    col_store<CH, M, CS>(a2, wave, vec, lane, s_slot + NWV * CS);
    __syncthreads();
    // per-channel totals -> parameter-gradient partials; gamma-weighted group sums for the second pass
    for (int t = tid; t < IPW * CH; t += SIR_THREADS) {
        const int im_ = t / CH, c = t % CH;
        const float A1 = col_total<IPW, CS>(s_slot, im_, c), A2 = col_total<IPW, CS>(s_slot + NWV * CS, im_, c);
        if (n0 + im_ < N) {
            part[((n0 + im_) * 3 + 0) * CH + c] = A2;
            part[((n0 + im_) * 3 + 1) * CH + c] = A1;
        }
        const float gm = gamma[c];
        float u = A1 * gm, v = A2 * gm;
#pragma unroll
        for (int o = 1; o < T::CPG; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
        if ((c % T::CPG) == 0) {
            const float inv = 1.0f / (float)(T::CPG * HW);
            s_grp[(im_ * SIR_G + c / T::CPG) * 2] = u * inv;
            s_grp[(im_ * SIR_G + c / T::CPG) * 2 + 1] = v * inv;
        }
    }
    __syncthreads();
    float db[8], S1[8], S2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int grp = (vec * 8 + j) / T::CPG;
        db[j] = 0.f;
        S1[j] = s_grp[(img * SIR_G + grp) * 2]; S2[j] = s_grp[(img * SIR_G + grp) * 2 + 1];
    }
#pragma unroll
    for (int k = 0; k < T::NV; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)raw[k][j] - mean[j]) * rstd[j];
            const float d = rstd[j] * (ga[j] * g[k][j] - (S1[j] + xh * S2[j]));
            g[k][j] = d;
            db[j] += d;
        }
    }
    col_store<CH, M, CS>(db, wave, vec, lane, s_slot + 2 * NWV * CS);
}

// After a barrier that follows norm_bwd: row 2 of the partial table (column sums of the raw gradient).
template <int CH, int IPW, int CS>
__device__ __forceinline__ void col2_flush(const float* s_slot, float* part, long long n0, int N, int tid) {
    constexpr int NWV = SIR_THREADS / 64;
    for (int t = tid; t < IPW * CH; t += SIR_THREADS) {
        const int img = t / CH, c = t % CH;
        const float v = col_total<IPW, CS>(s_slot + 2 * NWV * CS, img, c);
        if (n0 + img < N) part[((n0 + img) * 3 + 2) * CH + c] = v;
    }
}

template <int C, int W, int ACT>
__device__ __forceinline__ void sir_bwd_body(const SirBwdArgs& a, char* smem) {
    constexpr int E = 2 * C, HW = W * W, IPW = sir_ipw(HW), M = IPW * HW, MR = sir_tile_rows(M);
    constexpr int XS = C * 2 + 16, ES = E * 2 + 16;
    constexpr int NWV = SIR_THREADS / 64, MT = MR / 32;
    using E1 = Ew<E, M>;
    using E3 = Ew<C, M>;
    char* s_x = smem;                              // [M][XS]  d_o, later the expansion data-gradient
    char* s_e = s_x + MR * XS;                     // [M][ES]  projection data-gradient -> dd -> d_e
    float* s_col = (float*)(s_e + MR * ES);        // [3][waves][E] per-wave column sums
    float* s_grp = s_col + 3 * NWV * E;            // [IPW][16][2]
    float* s_mr = s_grp + IPW * 2 * SIR_G;         // [3][IPW][16][2]  mean, rstd of the three norms
    float* s_p = s_mr + 3 * IPW * 2 * SIR_G;       // per-channel parameters
    constexpr int P_G1 = 0, P_B1 = E, P_G2 = 2 * E, P_B2 = 3 * E, P_WD = 4 * E, P_G3 = 13 * E;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long n0 = (long long)blockIdx.x * IPW;
    const long long row0 = n0 * HW;
    const int rows_valid = IPW == 1 ? M : (int)min((long long)M, (a.N - n0) * HW);
    auto live = [&](int row) { return IPW == 1 || row < rows_valid; };      // one image per workgroup: never ragged, the row checks fold away
    const bool use_mask = a.m_e != nullptr;
    // workgroup-uniform bases of this workgroup's rows; lanes add 32-bit byte offsets (ld_at / st_at)
    const bf16* const dyb = a.dy + row0 * C; const bf16* const raw_ob = a.raw_o + row0 * C;
    const bf16* const raw_eb = a.raw_e + row0 * E; const bf16* const raw_db = a.raw_d + row0 * E; const bf16* const y_eb = a.y_e + row0 * E;
    const unsigned char* const m_inb = a.m_in + (row0 * C >> 3);
    const unsigned char* const m_eb = a.m_e + (row0 * E >> 3); const unsigned char* const m_db = a.m_d + (row0 * E >> 3);
    bf16* const dxb = a.dx + row0 * C; bf16* const d_ob = a.d_o + row0 * C; bf16* const d_eb = a.d_e + row0 * E;

    SIRB_STAMP(0);
    // ---- one round trip for everything small
    for (int i = tid; i < 9 * E; i += SIR_THREADS) s_p[P_WD + i] = a.wdf[i];
    for (int i = tid; i < E; i += SIR_THREADS) {
        s_p[P_G1 + i] = a.g1[i]; s_p[P_B1 + i] = a.b1[i]; s_p[P_G2 + i] = a.g2[i]; s_p[P_B2 + i] = a.b2[i];
    }
    for (int i = tid; i < C; i += SIR_THREADS) s_p[P_G3 + i] = a.g3[i];
    for (int i = tid; i < 3 * IPW * 2 * SIR_G; i += SIR_THREADS) {
        const int which = i / (IPW * 2 * SIR_G), rem = i % (IPW * 2 * SIR_G);
        const float* st = which == 0 ? a.st1 : which == 1 ? a.st2 : a.st3;
        s_mr[i] = (n0 + rem / (2 * SIR_G) < a.N) ? st[n0 * 2 * SIR_G + rem] : 0.f;
    }
    const int v1 = tid % E1::VPR, r1 = tid / E1::VPR;
    const int v3a = tid % E3::VPR, r3a = tid / E3::VPR;
    bf16x8 dyv[E3::NV], rawo[E3::NV];
    unsigned mk_none3[E3::NV];
#pragma unroll
    for (int k = 0; k < E3::NV; ++k) {
        const int row = r3a * E3::NV + k;
        const unsigned e = (unsigned)(row * C + v3a * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { dyv[k][j] = (bf16)0.f; rawo[k][j] = (bf16)0.f; }
        mk_none3[k] = 0xFFu;
        if (live(row)) {
            dyv[k] = ld_at<bf16x8>(dyb, e * 2);
            rawo[k] = ld_at<bf16x8>(raw_ob, e * 2);
        }
    }
    bf16x8 rawd[E1::NV];
    unsigned mk_e[E1::NV], mk_d[E1::NV];
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) {
        const int row = r1 * E1::NV + k;
        const unsigned e = (unsigned)(row * E + v1 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) rawd[k][j] = (bf16)0.f;
        mk_e[k] = mk_d[k] = 0xFFu;
        if (live(row)) {
            rawd[k] = ld_at<bf16x8>(raw_db, e * 2);
            if (use_mask) mk_d[k] = ld_at<unsigned char>(m_db, e >> 3);
        }
    }
    bf16x8 bfrag1[GemmK<C>::KCH];
    if (wave < MT * (E / 32)) gemm_prefetch<C>(bfrag1, a.wpd + (long long)((wave / MT) * 32 + (lane & 31)) * C, lane);
    __syncthreads();
    SIRB_STAMP(1);

    // ---- last GroupNorm backward (no activation, no dropout): dy -> d_o
    {
        float g[E3::NV][8];
#pragma unroll
        for (int k = 0; k < E3::NV; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) g[k][j] = (float)dyv[k][j];
        norm_bwd<C, M, HW, IPW, E, ACT_NONE>(rawo, g, mk_none3, false, 1.f, s_p + P_G3, s_p + P_G3, s_mr + 2 * IPW * 2 * SIR_G, s_col, s_grp,
                                            a.part3, n0, a.N, tid);
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const int row = r3a * E3::NV + k;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)g[k][j];
            *(bf16x8*)(s_x + row * XS + v3a * 16) = o;
            if (live(row)) st_at(d_ob, (unsigned)(row * C + v3a * 8) * 2, o);
        }
    }
    __syncthreads();
    col2_flush<C, IPW, E>(s_col, a.part3, n0, a.N, tid);
    SIRB_STAMP(2);

    // ---- projection data-gradient: [M][C] x [E][C]^T -> LDS tile (bf16)
    {
        constexpr int TT = MT * (E / 32);
#pragma unroll 1
        for (int t = wave; t < TT; t += NWV) {
            const int mt = t % MT, nt = t / MT;
            const int col = nt * 32 + (lane & 31);
            if (t != wave) gemm_prefetch<C>(bfrag1, a.wpd + (long long)col * C, lane);
            const f32x16 acc = gemm_tile<C, XS>(s_x, mt * 32, a.wpd + (long long)col * C, lane, bfrag1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_e + row * ES + col * 2) = (bf16)acc[r];
            }
        }
    }
    __syncthreads();

    SIRB_STAMP(3);
    // ---- second GroupNorm + activation + dropout backward -> dd (kept in registers as fp32 and in the LDS tile as bf16)
    float g[E1::NV][8];
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) {
        const bf16x8 t = *(const bf16x8*)(s_e + (r1 * E1::NV + k) * ES + v1 * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[k][j] = (float)t[j];
    }
    norm_bwd<E, M, HW, IPW, E, ACT>(rawd, g, mk_d, use_mask, a.mscale, s_p + P_G2, s_p + P_B2, s_mr + IPW * 2 * SIR_G, s_col, s_grp, a.part2,
                                   n0, a.N, tid, ACT_HAS_BETA(ACT) ? *a.ab2 : 1.f, ACT_HAS_BETA(ACT) ? a.dbp2 + blockIdx.x * NWV : nullptr);
    // the forward activation feeding the depthwise conv and the raw expansion output are needed next: request them now
    // (the widest tile has no registers to spare for the raw values during the depthwise phase: it asks for them afterwards)
    constexpr bool RAW_E_LATE = E1::NV * E >= 1024;
    bf16x8 yev[E1::NV], rawe[E1::NV];
    auto load_raw_e = [&](int r1, int v1) {
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) {
            const int row = r1 * E1::NV + k;
            const unsigned e = (unsigned)(row * E + v1 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) rawe[k][j] = (bf16)0.f;
            if (live(row)) {
                rawe[k] = ld_at<bf16x8>(raw_eb, e * 2);
                if (use_mask) mk_e[k] = ld_at<unsigned char>(m_eb, e >> 3);
            }
        }
    };
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) {
        const int row = r1 * E1::NV + k;
#pragma unroll
        for (int j = 0; j < 8; ++j) yev[k][j] = (bf16)0.f;
        if (live(row)) yev[k] = ld_at<bf16x8>(y_eb, (unsigned)(row * E + v1 * 8) * 2);
    }
    if constexpr (!RAW_E_LATE) load_raw_e(r1, v1);
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)g[k][j];
        *(bf16x8*)(s_e + (r1 * E1::NV + k) * ES + v1 * 16) = o;
    }
    __syncthreads();
    // table 2 of this norm = depthwise bias gradient: all waves (= all images of the workgroup) folded into row 9
    float* dwp = a.dwpart + (long long)blockIdx.x * 10 * E;
    for (int c = tid; c < E; c += SIR_THREADS) dwp[9 * E + c] = col_total<1, E>(s_col + 2 * NWV * E, 0, c);

    SIRB_STAMP(4);
    // ---- depthwise: data-gradient (flipped taps) and weight-gradient share the neighbour reads of dd
    {
        float acc[E1::NV][8];
#pragma unroll
        for (int k = 0; k < E1::NV; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // neighbour q - off(tap) of the forward stencil = q + off(8 - tap); wdf holds the taps flipped, so tap index
            // t2 = 8 - tap of wdf multiplies dd[q + off(t2)] in the data-gradient, and the same neighbour times the
            // forward activation at q is the gradient of forward tap `tap`.
            const int t2 = 8 - tap;
            const int dy = t2 / 3 - 1, dx = t2 % 3 - 1;
            float w[8], wg[8];
            const f32x4 w0 = *(const f32x4*)(s_p + P_WD + t2 * E + v1 * 8), w1 = *(const f32x4*)(s_p + P_WD + t2 * E + v1 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = w0[j]; w[4 + j] = w1[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) wg[j] = 0.f;
#pragma unroll
            for (int k = 0; k < E1::NV; ++k) {
                const int row = r1 * E1::NV + k;
                const int p = row % HW, py = p / W, px = p % W;
                const int ny = py + dy, nx = px + dx;
                if (ny >= 0 && ny < W && nx >= 0 && nx < W) {
                    const bf16x8 nv = *(const bf16x8*)(s_e + (row + dy * W + dx) * ES + v1 * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = (float)nv[j];
                        acc[k][j] += f * w[j];
                        wg[j] += f * (float)yev[k][j];
                    }
                }
            }
            // tap gradient of the workgroup: per-wave sums into table (tap & 1) (tables 0 and 1 are idle between the two
            // norms), then the thread that owns a channel adds the waves in a fixed order.  The barrier of the next
            // tap separates these reads from the next writes of the same table.
            float* tab = s_col + (tap & 1) * NWV * E;
            col_store<E, M, E>(wg, wave, v1, lane, tab);
            __syncthreads();
            for (int c = tid; c < E; c += SIR_THREADS) dwp[tap * E + c] = col_total<1, E>(tab, 0, c);
            // pins the accumulators between taps: left alone, the scheduler interleaves the taps' neighbour reads and spills
#pragma unroll
            for (int k = 0; k < E1::NV; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[k][j]));
        }
#pragma unroll
        for (int k = 0; k < E1::NV; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) g[k][j] = (float)(bf16)acc[k][j];     // the unfused path stores this tensor as bf16
    }
    __syncthreads();                                                          // the last tap's table is read before the next norm refills it

    SIRB_STAMP(5);
    // ---- first GroupNorm + activation + dropout backward -> d_e
    // (thread indices formed again from an opaque copy: the addresses of this phase are then computed here, not carried -
    // in registers the wide tiles do not have - from the phase above that used the same ones)
    int tid1 = tid;
    asm volatile("" : "+v"(tid1));
    const int v1b = tid1 % E1::VPR, r1b = tid1 / E1::VPR;
    if constexpr (RAW_E_LATE) load_raw_e(r1b, v1b);
    bf16x8 bfrag2[GemmK<E>::KCH];
    if (wave < MT * (C / 32)) gemm_prefetch<E>(bfrag2, a.wed + (long long)((wave / MT) * 32 + (lane & 31)) * E, lane);
    norm_bwd<E, M, HW, IPW, E, ACT>(rawe, g, mk_e, use_mask, a.mscale, s_p + P_G1, s_p + P_B1, s_mr, s_col + 0, s_grp, a.part1, n0, a.N, tid1,
                                   ACT_HAS_BETA(ACT) ? *a.ab1 : 1.f, ACT_HAS_BETA(ACT) ? a.dbp1 + blockIdx.x * NWV : nullptr);
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) {
        const int row = r1b * E1::NV + k;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)g[k][j];
        *(bf16x8*)(s_e + row * ES + v1b * 16) = o;
        if (live(row)) st_at(d_eb, (unsigned)(row * E + v1b * 8) * 2, o);
    }
    __syncthreads();
    col2_flush<E, IPW, E>(s_col, a.part1, n0, a.N, tid1);

    SIRB_STAMP(6);
    // ---- expansion data-gradient: [M][E] x [C][E]^T -> LDS (bf16)
    {
        constexpr int TT = MT * (C / 32);
#pragma unroll 1
        for (int t = wave; t < TT; t += NWV) {
            const int mt = t % MT, nt = t / MT;
            const int col = nt * 32 + (lane & 31);
            if (t != wave) gemm_prefetch<E>(bfrag2, a.wed + (long long)col * E, lane);
            const f32x16 acc = gemm_tile<E, ES>(s_e, mt * 32, a.wed + (long long)col * E, lane, bfrag2);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_x + row * XS + col * 2) = (bf16)acc[r];
            }
        }
    }
    // the incoming gradient and the input's dropout bits again (L2 hits, in flight across the barrier): cheaper than
    // holding them in registers through every phase above
    bf16x8 dy2[E3::NV];
    unsigned mk_in[E3::NV];
    int tid_end = tid;
    asm volatile("" : "+v"(tid_end));              // the row / vector indices are formed again here rather than kept (or spilled) since the first phase
    const int v3 = tid_end % E3::VPR, r3 = tid_end / E3::VPR;
#pragma unroll
    for (int k = 0; k < E3::NV; ++k) {
        const int row = r3 * E3::NV + k;
        const unsigned e = (unsigned)(row * C + v3 * 8);
        mk_in[k] = 0xFFu;
        if (live(row)) {
            dy2[k] = ld_at<bf16x8>(dyb, e * 2);
            if (a.m_in) mk_in[k] = ld_at<unsigned char>(m_inb, e >> 3);
        }
    }
    __syncthreads();

    SIRB_STAMP(7);
    // ---- dropout of the block input and the residual branch: dx = dy + mask * scale * (expansion data-gradient)
#pragma unroll
    for (int k = 0; k < E3::NV; ++k) {
        const int row = r3 * E3::NV + k;
        if (!live(row)) continue;
        const bf16x8 t = *(const bf16x8*)(s_x + row * XS + v3 * 16);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = (float)t[j];
            if (a.m_in) d = (mk_in[k] >> j) & 1 ? (float)(bf16)(d * a.mscale) : 0.f;
            o[j] = (bf16)(d + (float)dy2[k][j]);
        }
        st_at(dxb, (unsigned)(row * C + v3 * 8) * 2, o);
    }
    SIRB_STAMP(8);
}


// GroupNorm (+activation, +dropout) backward for a tile too large to keep an fp32 copy of the gradient in registers:
// the first pass reads the incoming gradient from the LDS tile (bf16) and the raw tensor from global memory, evaluates
// the activation derivative and puts g = dy * mask * act'(z) back into the tile (bf16, every thread owns its elements);
// the second pass reads g from the tile and the raw tensor again (an L2 hit), so the derivative - most of the
// instructions of this VALU-bound phase - is evaluated once.  The result replaces g in the tile and optionally goes to
// global memory.  Same slot protocol as norm_bwd.
template <int CH, int M, int HW, int CS, int ACT>
__device__ __forceinline__ void norm_bwd_tile(char* tile, int TS, const bf16* raw_g, const unsigned char* mask_g, long long elem0, float mscale,
                                              const float* gamma, const float* beta, const float* s_mr, float* s_slot, float* s_grp,
                                              float* part, bf16* out_g, long long n0, int tid, float bsc = 1.f, float* dbp = nullptr) {
    using T = Ew<CH, M>;
    constexpr int NWV = SIR_THREADS / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int vec = tid % T::VPR, r0 = tid / T::VPR;
    float dbs = 0.f;                               // swish / hardswish: this thread's share of the scalar's gradient
    float ga[8], be[8], mean[8], rstd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int grp = (vec * 8 + j) / T::CPG;
        ga[j] = gamma[vec * 8 + j]; be[j] = ACT != ACT_NONE ? beta[vec * 8 + j] : 0.f;
        mean[j] = s_mr[grp * 2]; rstd[j] = s_mr[grp * 2 + 1];
    }
    // gradient w.r.t. the normalised (pre-activation) value and xhat, for 4 vectors of this thread
    auto chunk = [&](int k0, float (&gg)[4][8], float (&xh)[4][8]) {
        bf16x8 raw[4], gin[4];
        unsigned mk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = r0 * T::NV + k0 + k;
            const long long e = (long long)row * CH + vec * 8;
            raw[k] = *(const bf16x8*)(raw_g + e);
            mk[k] = mask_g ? mask_g[(elem0 + e) >> 3] : 0xFFu;
            gin[k] = *(const bf16x8*)(tile + row * TS + vec * 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[k][j] = ((float)raw[k][j] - mean[j]) * rstd[j];
                float v = (float)gin[k][j];
                if constexpr (ACT_HAS_BETA(ACT)) {
                    float db;
                    const float d = act_d_db<ACT>(xh[k][j] * ga[j] + be[j], bsc, db);
                    db *= v; v *= d;
                    if (mask_g) db = (mk[k] >> j) & 1 ? db * mscale : 0.f;
                    dbs += db;
                } else if constexpr (ACT != ACT_NONE) {
                    v *= act_d<ACT>(xh[k][j] * ga[j] + be[j]);
                }
                if (mask_g) v = (mk[k] >> j) & 1 ? v * mscale : 0.f;
                gg[k][j] = v;
            }
    };
    // second pass: g comes back from the tile, only xhat is recomputed
    auto chunk2 = [&](int k0, float (&gg)[4][8], float (&xh)[4][8]) {
        bf16x8 raw[4], gin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = r0 * T::NV + k0 + k;
            raw[k] = *(const bf16x8*)(raw_g + (long long)row * CH + vec * 8);
            gin[k] = *(const bf16x8*)(tile + row * TS + vec * 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[k][j] = ((float)raw[k][j] - mean[j]) * rstd[j];
                gg[k][j] = (float)gin[k][j];
            }
    };
    float a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll 1
    for (int k0 = 0; k0 < T::NV; k0 += 4) {
        float gg[4][8], xh[4][8];
        chunk(k0, gg, xh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (bf16)gg[k][j];
                const float gr = (float)o[j];            // the sums see the value the second pass will see
                a1[j] += gr; a2[j] += gr * xh[k][j];
            }
            *(bf16x8*)(tile + (r0 * T::NV + k0 + k) * TS + vec * 16) = o;
        }
    }
    if constexpr (ACT_HAS_BETA(ACT)) {
        dbs = wave_sum(dbs);
        if (lane == 0) dbp[wave] = dbs;
    }
    col_store<CH, M, CS>(a1, wave, vec, lane, s_slot);
    col_store<CH, M, CS>(a2, wave, vec, lane, s_slot + NWV * CS);
    __syncthreads();
    for (int t = tid; t < CH; t += SIR_THREADS) {
        const float A1 = col_total<1, CS>(s_slot, 0, t), A2 = col_total<1, CS>(s_slot + NWV * CS, 0, t);
        part[(n0 * 3 + 0) * CH + t] = A2;
        part[(n0 * 3 + 1) * CH + t] = A1;
        const float gm = gamma[t];
        float u = A1 * gm, v = A2 * gm;
#pragma unroll
        for (int o = 1; o < T::CPG; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
        if ((t % T::CPG) == 0) {
            const float inv = 1.0f / (float)(T::CPG * HW);
            s_grp[(t / T::CPG) * 2] = u * inv;
            s_grp[(t / T::CPG) * 2 + 1] = v * inv;
        }
    }
    __syncthreads();
    float db[8], S1[8], S2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int grp = (vec * 8 + j) / T::CPG;
        db[j] = 0.f;
        S1[j] = s_grp[grp * 2]; S2[j] = s_grp[grp * 2 + 1];
    }
#pragma unroll 1
    for (int k0 = 0; k0 < T::NV; k0 += 4) {
        float gg[4][8], xh[4][8];
        chunk2(k0, gg, xh);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = r0 * T::NV + k0 + k;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = rstd[j] * (ga[j] * gg[k][j] - (S1[j] + xh[k][j] * S2[j]));
                db[j] += d;
                o[j] = (bf16)d;
            }
            *(bf16x8*)(tile + row * TS + vec * 16) = o;
            if (out_g) *(bf16x8*)(out_g + (long long)row * CH + vec * 8) = o;
        }
    }
    col_store<CH, M, CS>(db, wave, vec, lane, s_slot + 2 * NWV * CS);
}

// Backward for images whose expanded tile fills the LDS on its own (32x32 pixels at C = 32): one [M][E] tile that is also
// used, as an [M][C] view, for d_o before the projection data-gradient and for the expansion data-gradient at the end;
// GEMM results that would overwrite their own operand stay in accumulators across a barrier.
template <int C, int W, int ACT>
__device__ __forceinline__ void sir_bwd_big_body(const SirBwdArgs& a, char* smem) {
    constexpr int E = 2 * C, HW = W * W, M = HW;
    constexpr int XS = C * 2 + 16, ES = E * 2 + 16;
    constexpr int NWV = SIR_THREADS / 64, MT = M / 32, TPW = MT / NWV;
    static_assert(C == 32 && MT % NWV == 0, "big variant: one input-channel tile, M tiles dealt evenly");
    using E1 = Ew<E, M>;
    using E3 = Ew<C, M>;
    char* s_e = smem;
    float* s_col = (float*)(smem + M * ES);        // [3][waves][E]
    float* s_grp = s_col + 3 * NWV * E;            // [16][2]
    float* s_mr = s_grp + 2 * SIR_G;               // [3][16][2]
    float* s_p = s_mr + 3 * 2 * SIR_G;
    constexpr int P_G1 = 0, P_B1 = E, P_G2 = 2 * E, P_B2 = 3 * E, P_WD = 4 * E, P_G3 = 13 * E;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long n0 = blockIdx.x, row0 = n0 * HW;
    const bool use_mask = a.m_e != nullptr;

    for (int i = tid; i < 9 * E; i += SIR_THREADS) s_p[P_WD + i] = a.wdf[i];
    for (int i = tid; i < E; i += SIR_THREADS) {
        s_p[P_G1 + i] = a.g1[i]; s_p[P_B1 + i] = a.b1[i]; s_p[P_G2 + i] = a.g2[i]; s_p[P_B2 + i] = a.b2[i];
    }
    for (int i = tid; i < C; i += SIR_THREADS) s_p[P_G3 + i] = a.g3[i];
    for (int i = tid; i < 3 * 2 * SIR_G; i += SIR_THREADS) {
        const int which = i / (2 * SIR_G), rem = i % (2 * SIR_G);
        const float* st = which == 0 ? a.st1 : which == 1 ? a.st2 : a.st3;
        s_mr[i] = st[n0 * 2 * SIR_G + rem];
    }
    const int v1 = tid % E1::VPR, r1 = tid / E1::VPR;
    const int v3 = tid % E3::VPR, r3 = tid / E3::VPR;
    __syncthreads();
    SIRB_STAMP(0);

    // ---- last GroupNorm backward: dy -> d_o (LDS [M][C] view + global)
    {
        bf16x8 rawo[E3::NV];
        float g[E3::NV][8];
        unsigned none[E3::NV];
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const long long e = (row0 + r3 * E3::NV + k) * C + v3 * 8;
            const bf16x8 d = *(const bf16x8*)(a.dy + e);
            rawo[k] = *(const bf16x8*)(a.raw_o + e);
            none[k] = 0xFFu;
#pragma unroll
            for (int j = 0; j < 8; ++j) g[k][j] = (float)d[j];
        }
        norm_bwd<C, M, HW, 1, E, ACT_NONE>(rawo, g, none, false, 1.f, s_p + P_G3, s_p + P_G3, s_mr + 2 * 2 * SIR_G, s_col, s_grp, a.part3,
                                          n0, a.N, tid);
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const int row = r3 * E3::NV + k;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)g[k][j];
            *(bf16x8*)(s_e + row * XS + v3 * 16) = o;
            *(bf16x8*)(a.d_o + (row0 + row) * C + v3 * 8) = o;
        }
    }
    __syncthreads();
    col2_flush<C, 1, E>(s_col, a.part3, n0, a.N, tid);
    SIRB_STAMP(1);

    // ---- projection data-gradient: results wait in accumulators until every wave has read d_o, then fill the [M][E] tile
    {
        f32x16 acc[TPW][E / 32];
        bf16x8 bfr[E / 32][C / 16];
#pragma unroll
        for (int nt = 0; nt < E / 32; ++nt)
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks)
                bfr[nt][ks] = *(const bf16x8*)(a.wpd + (long long)(nt * 32 + (lane & 31)) * C + ks * 16 + (lane >> 5) * 8);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const char* ap = s_e + ((wave + NWV * i) * 32 + (lane & 31)) * XS + (lane >> 5) * 16;
            bf16x8 av[C / 16];
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks) av[ks] = *(const bf16x8*)(ap + ks * 32);
#pragma unroll
            for (int nt = 0; nt < E / 32; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][nt][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < C / 16; ++ks) acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ks], bfr[nt][ks], acc[i][nt], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int nt = 0; nt < E / 32; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave + NWV * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    *(bf16*)(s_e + row * ES + (nt * 32 + (lane & 31)) * 2) = (bf16)acc[i][nt][r];
                }
    }
    __syncthreads();
    SIRB_STAMP(2);

    // ---- second GroupNorm + activation + dropout backward -> dd (in place in the tile)
    norm_bwd_tile<E, M, HW, E, ACT>(s_e, ES, a.raw_d + row0 * E, use_mask ? a.m_d : nullptr, row0 * E, a.mscale, s_p + P_G2, s_p + P_B2,
                                    s_mr + 2 * SIR_G, s_col, s_grp, a.part2, nullptr, n0, tid, ACT_HAS_BETA(ACT) ? *a.ab2 : 1.f,
                                    ACT_HAS_BETA(ACT) ? a.dbp2 + blockIdx.x * NWV : nullptr);
    __syncthreads();
    SIRB_STAMP(3);
    float* dwp = a.dwpart + n0 * 10 * E;
    for (int c = tid; c < E; c += SIR_THREADS) dwp[9 * E + c] = col_total<1, E>(s_col + 2 * NWV * E, 0, c);

    // ---- depthwise data-gradient and weight-gradient, four vectors at a time; the results replace dd in the tile once
    //      every thread has finished reading its neighbours
    bf16x8 dea[E1::NV];
    static_assert(E1::VPR == 8, "a lane keeps one folded tap sum");
    float tapsum[9];                               // this lane's share of the wave's tap gradients: channel v1 * 8 + tapj
    int tapj = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) tapsum[t] = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < E1::NV; k0 += 4) {
        float acc[4][8];
        bf16x8 yev[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            yev[k] = *(const bf16x8*)(a.y_e + (row0 + r1 * E1::NV + k0 + k) * E + v1 * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int t2 = 8 - tap;
            const int dy = t2 / 3 - 1, dx = t2 % 3 - 1;
            float w[8], wg[8];
            const f32x4 w0 = *(const f32x4*)(s_p + P_WD + t2 * E + v1 * 8), w1 = *(const f32x4*)(s_p + P_WD + t2 * E + v1 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = w0[j]; w[4 + j] = w1[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) wg[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r1 * E1::NV + k0 + k;
                const int py = row / W, px = row % W;
                const int ny = py + dy, nx = px + dx;
                if (ny >= 0 && ny < W && nx >= 0 && nx < W) {
                    const bf16x8 nv = *(const bf16x8*)(s_e + (row + dy * W + dx) * ES + v1 * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float f = (float)nv[j];
                        acc[k][j] += f * w[j];
                        wg[j] += f * (float)yev[k][j];
                    }
                }
            }
            tapj = fold_cols<E1::VPR>(wg, lane);
            tapsum[tap] += wg[0];
            asm volatile("" : "+v"(tapsum[tap]));          // (with the one below) pins the order of evaluation: left alone the scheduler interleaves the chunks and spills
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) dea[k0 + k][j] = (bf16)acc[k][j];
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(dea[k0 + k]));    // convert now: the fp32 accumulators of a finished chunk are dead
    }
    SIRB_STAMP(4);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) *(bf16x8*)(s_e + (r1 * E1::NV + k) * ES + v1 * 16) = dea[k];
    // tap gradients of the workgroup, three taps per round through the three (now idle) column tables: per-wave sums,
    // then the thread that owns (tap, channel) adds the waves in a fixed order
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
#pragma unroll
        for (int t = 0; t < 3; ++t) s_col[(t * NWV + wave) * E + v1 * 8 + tapj] = tapsum[t0 + t];
        __syncthreads();
        if (tid < 3 * E) dwp[t0 * E + tid] = col_total<1, E>(s_col + (tid / E) * NWV * E, 0, tid % E);
        __syncthreads();
    }

    SIRB_STAMP(5);
    // ---- first GroupNorm + activation + dropout backward -> d_e (tile + global); own elements only, no barrier needed before
    norm_bwd_tile<E, M, HW, E, ACT>(s_e, ES, a.raw_e + row0 * E, use_mask ? a.m_e : nullptr, row0 * E, a.mscale, s_p + P_G1, s_p + P_B1, s_mr,
                                    s_col, s_grp, a.part1, a.d_e + row0 * E, n0, tid, ACT_HAS_BETA(ACT) ? *a.ab1 : 1.f,
                                    ACT_HAS_BETA(ACT) ? a.dbp1 + blockIdx.x * NWV : nullptr);
    __syncthreads();
    col2_flush<E, 1, E>(s_col, a.part1, n0, a.N, tid);
    SIRB_STAMP(6);

    // ---- expansion data-gradient (accumulators across a barrier, then the [M][C] view of the tile)
    {
        f32x16 acc[TPW];
        bf16x8 bfr[E / 16];
#pragma unroll
        for (int ks = 0; ks < E / 16; ++ks) bfr[ks] = *(const bf16x8*)(a.wed + (long long)(lane & 31) * E + ks * 16 + (lane >> 5) * 8);
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            const char* ap = s_e + ((wave + NWV * i) * 32 + (lane & 31)) * ES + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < E / 16; ++ks) {
                const bf16x8 av = *(const bf16x8*)(ap + ks * 32);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bfr[ks], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wave + NWV * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_e + row * XS + (lane & 31) * 2) = (bf16)acc[i][r];
            }
    }
    __syncthreads();
    SIRB_STAMP(7);

    // ---- dropout of the block input and the residual branch
#pragma unroll
    for (int k = 0; k < E3::NV; ++k) {
        const int row = r3 * E3::NV + k;
        const long long e = (row0 + row) * C + v3 * 8;
        const bf16x8 t = *(const bf16x8*)(s_e + row * XS + v3 * 16);
        const bf16x8 dyv = *(const bf16x8*)(a.dy + e);
        const unsigned m = a.m_in ? a.m_in[e >> 3] : 0xFFu;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = (float)t[j];
            if (a.m_in) d = (m >> j) & 1 ? (float)(bf16)(d * a.mscale) : 0.f;
            o[j] = (bf16)(d + (float)dyv[j]);
        }
        *(bf16x8*)(a.dx + e) = o;
    }
    SIRB_STAMP(8);
}

template <int C, int W>
__global__ __launch_bounds__(SIR_THREADS) void sir_bwd_big_kernel(const SirBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    switch (a.act) {
        case ACT_GELU: sir_bwd_big_body<C, W, ACT_GELU>(a, smem); break;
        case ACT_SILU: sir_bwd_big_body<C, W, ACT_SILU>(a, smem); break;
        case ACT_SWISH: sir_bwd_big_body<C, W, ACT_SWISH>(a, smem); break;
        case ACT_HARDSWISH: sir_bwd_big_body<C, W, ACT_HARDSWISH>(a, smem); break;
        default: sir_bwd_big_body<C, W, ACT_RELU>(a, smem); break;
    }
}

template <int C, int W>
static int sir_bwd_big_launch(const SirBwdArgs& a, hipStream_t stream) {
    constexpr int E = 2 * C, M = W * W;
    constexpr int lds = M * (E * 2 + 16) + (3 * (SIR_THREADS / 64) * E + 2 * SIR_G + 3 * 2 * SIR_G + 13 * E + C) * 4;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    auto k = sir_bwd_big_kernel<C, W>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            ddpm_set_error("sir_backward: cannot reserve %d bytes of LDS", lds);
            return DDPM_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(a.N), dim3(SIR_THREADS), lds, stream, a);
    return ddpm_check_launch("sir_backward");
}

// The activation is a template argument of the kernel (not a switch inside it): each activation's body is then
// register-allocated on its own, and the widest shape needs no scratch memory.
// Waves per SIMD the register allocation must leave room for.  The narrow tiles land within a few registers of 128, the
// limit under which two workgroups share a CU: with GELU (the reference's default) they are held to it, which costs no
// scratch memory; the other activations keep their natural allocation (122 - 134 registers).
constexpr int sir_bwd_waves(int c, int w, int act) { return (c * w <= 512 && w <= 8 && act == ACT_GELU) ? 4 : 2; }

template <int C, int W, int ACT>
__global__ __launch_bounds__(SIR_THREADS, sir_bwd_waves(C, W, ACT)) void sir_bwd_kernel(const SirBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sir_bwd_body<C, W, ACT>(a, smem);
}

template <int C, int W, int ACT>
static int sir_bwd_launch_act(const SirBwdArgs& a, hipStream_t stream) {
    constexpr int E = 2 * C, HW = W * W, IPW = sir_ipw(HW), M = sir_tile_rows(IPW * HW);
    constexpr int lds = M * (C * 2 + 16) + M * (E * 2 + 16) +
                        (3 * (SIR_THREADS / 64) * E + IPW * 2 * SIR_G + 3 * IPW * 2 * SIR_G + 13 * E + C) * 4;
    static_assert(lds <= 160 * 1024, "tiles do not fit LDS");
    auto k = sir_bwd_kernel<C, W, ACT>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            ddpm_set_error("sir_backward: cannot reserve %d bytes of LDS", lds);
            return DDPM_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3((a.N + IPW - 1) / IPW), dim3(SIR_THREADS), lds, stream, a);
    return ddpm_check_launch("sir_backward");
}

template <int C, int W>
static int sir_bwd_launch(const SirBwdArgs& a, hipStream_t stream) {
    switch (a.act) {
        case ACT_GELU: return sir_bwd_launch_act<C, W, ACT_GELU>(a, stream);
        case ACT_SILU: return sir_bwd_launch_act<C, W, ACT_SILU>(a, stream);
        case ACT_SWISH:
        case ACT_HARDSWISH:
            // (the two widest tiles have no registers left for the scalar's gradient: ddpm_sir_backward_supported says so and
            // their backward runs on the per-unit kernels)
            if constexpr (sir_bwd_takes_scalar_acts(C, W)) {
                return a.act == ACT_SWISH ? sir_bwd_launch_act<C, W, ACT_SWISH>(a, stream) : sir_bwd_launch_act<C, W, ACT_HARDSWISH>(a, stream);
            } else {
                ddpm_set_error("sir_backward: swish / hardswish are not available for C=%d W=%d", C, W);
                return DDPM_EINVAL;
            }
        default: return sir_bwd_launch_act<C, W, ACT_RELU>(a, stream);
    }
}

extern "C" int ddpm_sir_workgroups(int HW, int N) {
    const int ipw = sir_ipw(HW);
    return (N + ipw - 1) / ipw;
}

extern "C" int ddpm_sir_backward(const ddpm_sir_bwd_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(d && d->dy && d->raw_e && d->raw_d && d->raw_o && d->y_e && d->st1 && d->st2 && d->st3 && d->g1 && d->b1 && d->g2 &&
                     d->b2 && d->g3 && d->wpd && d->wed && d->wdf && d->dx && d->d_o && d->d_e && d->part1 && d->part2 && d->part3 &&
                     d->dwpart, "sir_backward: null pointer");
    DDPM_REQUIRE(d->N > 0, "sir_backward: N must be positive");
    DDPM_REQUIRE(ddpm_sir_backward_supported(d->C, d->H, d->W, d->E, d->groups, d->act),
                 "sir_backward: unsupported shape C=%d %dx%d E=%d groups=%d act=%d (use the unfused kernels)", d->C, d->H, d->W, d->E,
                 d->groups, d->act);
    DDPM_REQUIRE((d->m_in != nullptr) == (d->m_e != nullptr) && (d->m_e != nullptr) == (d->m_d != nullptr),
                 "sir_backward: give all three dropout masks or none");
    DDPM_REQUIRE(!ACT_HAS_BETA(d->act) || (d->act_b1 && d->act_b2 && d->dbeta1 && d->dbeta2),
                 "sir_backward: swish / hardswish need the two scalars (act_b1, act_b2) and the workspaces of their gradients (dbeta1, dbeta2)");
    SirBwdArgs a;
    a.dy = (const bf16*)d->dy; a.raw_e = (const bf16*)d->raw_e; a.raw_d = (const bf16*)d->raw_d; a.raw_o = (const bf16*)d->raw_o;
    a.y_e = (const bf16*)d->y_e; a.st1 = d->st1; a.st2 = d->st2; a.st3 = d->st3;
    a.g1 = d->g1; a.b1 = d->b1; a.g2 = d->g2; a.b2 = d->b2; a.g3 = d->g3;
    a.wpd = (const bf16*)d->wpd; a.wed = (const bf16*)d->wed; a.wdf = d->wdf;
    a.m_in = d->m_in; a.m_e = d->m_e; a.m_d = d->m_d; a.mscale = d->mask_scale;
    a.dx = (bf16*)d->dx; a.d_o = (bf16*)d->d_o; a.d_e = (bf16*)d->d_e;
    a.part1 = d->part1; a.part2 = d->part2; a.part3 = d->part3; a.dwpart = d->dwpart;
    a.N = d->N; a.act = d->act;
    a.ab1 = d->act_b1; a.ab2 = d->act_b2; a.dbp1 = d->dbeta1; a.dbp2 = d->dbeta2;
    if (d->C == 32 && d->W == 32) return sir_bwd_big_launch<32, 32>(a, stream);
    if (d->C == 32 && d->W == 16) return sir_bwd_launch<32, 16>(a, stream);
    if (d->C == 64 && d->W == 8) return sir_bwd_launch<64, 8>(a, stream);
    if (d->C == 128 && d->W == 4) return sir_bwd_launch<128, 4>(a, stream);
    if (d->C == 256 && d->W == 2) return sir_bwd_launch<256, 2>(a, stream);
    if (d->C == 128 && d->W == 8) return sir_bwd_launch<128, 8>(a, stream);
    if (d->C == 64 && d->W == 16) return sir_bwd_launch<64, 16>(a, stream);
    return sir_bwd_launch<256, 4>(a, stream);
}
