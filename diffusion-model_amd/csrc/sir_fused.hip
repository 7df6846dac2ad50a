// One SeparableInvertResidual block (conv_block.py:57-136 of the reference) as ONE kernel:
//
//     x --dropout--> 1x1 expand (C -> E=2C) -> GroupNorm -> act -> dropout
//                 -> depthwise 3x3            -> GroupNorm -> act -> dropout
//                 -> 1x1 project (E -> C)     -> GroupNorm -> (+ x)
//
// A workgroup owns whole images (one image, or 32/HW images when an image has fewer than 32 pixels, so that the
// MFMA tile has 32 rows), which makes every GroupNorm statistic a workgroup-local reduction.  The activations of
// the block never leave the CU: the block input and the expanded tensor live in LDS, the two pointwise
// convolutions run on v_mfma_f32_32x32x16_bf16 with the weights streamed straight from L2 into registers (each
// weight fragment is used by exactly one wave per workgroup), the depthwise stencil reads its neighbours from LDS.
// Unfused, the same block is six launches with four HBM round trips of the expanded tensor; at these sizes
// (<= 256 pixels x <= 512 channels per image) each of those launches is latency bound.
//
// In training the kernel also writes what the backward pass needs (raw convolution outputs, normalised
// activations, statistics); in inference it writes only the block output.
#include "common.h"
#include "ddpm_hip.h"
#include "sir_tile.h"

struct SirArgs {
    const bf16* x;
    bf16* y;
    const bf16* we; const float* be; const float* g1; const float* b1;      // expand  [E][C] bf16
    const float* wd; const float* bd; const float* g2; const float* b2;     // depthwise [9][E] fp32
    const bf16* wp; const float* bp; const float* g3; const float* b3;      // project [C][E] bf16
    const unsigned char* m_in; const unsigned char* m_e; const unsigned char* m_d;
    float mscale;
    bf16* raw_e; bf16* y_e; float* st1;
    bf16* raw_d; bf16* y_d; float* st2;
    bf16* raw_o; float* st3;
    int N, act;
    float eps;
    const float* ab1; const float* ab2;      // swish / hardswish: the two activation modules' learnable scalars (else null)
};

// Diagnostic build only (-DSIR_STAMPS): lane 0 of every wave records the shader clock at phase boundaries into a
// buffer of its own; no output value depends on it and the product build contains none of this.
#ifdef SIR_STAMPS
__device__ unsigned long long* sir_dbg = nullptr;
extern "C" int ddpm_sir_set_dbg(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(sir_dbg), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#define SIR_STAMP(i)                                                                                       \
    do {                                                                                                   \
        if (sir_dbg && (threadIdx.x & 63) == 0)                                                            \
            sir_dbg[((unsigned long long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define SIR_STAMP(i)
#endif

// Group statistics without atomics (bit-reproducible): every wave reduces the vectors it holds (all rows of a wave
// belong to one image) and stores one partial per group into its own slot; the finalising threads add the slots of
// their image in a fixed order.  Slot = pass * waves + wave when a workgroup holds several images (passes then
// belong to different images), else just the wave (partials are accumulated over the passes first).
template <int CH, int M>
__device__ __forceinline__ void group_store(const float (&s)[8], const float (&ss)[8], int slot, int v, int lane, float* s_part) {
    using E = Ew<CH, M>;
    float a[E::GPV], b[E::GPV];
#pragma unroll
    for (int q = 0; q < E::GPV; ++q) {
        a[q] = b[q] = 0.f;
#pragma unroll
        for (int j = 0; j < 8 / E::GPV; ++j) { a[q] += s[q * (8 / E::GPV) + j]; b[q] += ss[q * (8 / E::GPV) + j]; }
    }
#pragma unroll
    for (int q = 0; q < E::GPV; ++q) {
#pragma unroll
        for (int o = 1; o < E::VPG; o <<= 1) { a[q] += __shfl_xor(a[q], o, 64); b[q] += __shfl_xor(b[q], o, 64); }
#pragma unroll
        for (int o = E::VPR; o < 64; o <<= 1) { a[q] += __shfl_xor(a[q], o, 64); b[q] += __shfl_xor(b[q], o, 64); }
    }
    if (lane < E::VPR && (v % E::VPG) == 0) {
#pragma unroll
        for (int q = 0; q < E::GPV; ++q) {
            const int g = (v * 8) / E::CPG + q;
            s_part[(slot * 2 + 0) * SIR_G + g] = a[q];
            s_part[(slot * 2 + 1) * SIR_G + g] = b[q];
        }
    }
}

// mean / rstd of every (image, group) of the workgroup from the slots; thread t < IPW*16 owns (image t/16, group t%16).
template <int CH, int M, int HW, int IPW>
__device__ __forceinline__ void group_finish(int tid, const float* s_part, float* s_mr, float eps, float* st_out, long long n0, int N) {
    using E = Ew<CH, M>;
    constexpr int NWV = SIR_THREADS / 64;
    if (tid < IPW * SIR_G) {
        const int img = tid / SIR_G, g = tid % SIR_G;
        float S = 0.f, SS = 0.f;
        if constexpr (IPW == 1) {
#pragma unroll
            for (int w = 0; w < NWV; ++w) { S += s_part[(w * 2) * SIR_G + g]; SS += s_part[(w * 2 + 1) * SIR_G + g]; }
        } else {
#pragma unroll
            for (int k = 0; k < E::NV; ++k)
#pragma unroll
                for (int w = 0; w < NWV; ++w) {
                    const int slot_img = (w * (64 / E::VPR) + k * E::RPP) / HW;
                    const float p = s_part[((k * NWV + w) * 2) * SIR_G + g], q = s_part[((k * NWV + w) * 2 + 1) * SIR_G + g];
                    if (slot_img == img) { S += p; SS += q; }
                }
        }
        const float im = 1.0f / (float)(E::CPG * HW);
        const float mean = S * im;
        const float var = fmaxf(SS * im - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        s_mr[tid * 2] = mean; s_mr[tid * 2 + 1] = rstd;
        if (st_out && n0 + img < N) {
            st_out[(n0 * SIR_G + tid) * 2] = mean;
            st_out[(n0 * SIR_G + tid) * 2 + 1] = rstd;
        }
    }
}

template <int C, int W, int ACT>
__device__ __forceinline__ void sir_fwd_body(const SirArgs& a, char* smem) {
    constexpr int E = 2 * C, HW = W * W, IPW = sir_ipw(HW), M = IPW * HW, MR = sir_tile_rows(M);
    constexpr int XS = C * 2 + 16, ES = E * 2 + 16;
    constexpr int NWV = SIR_THREADS / 64;
    using E1 = Ew<E, M>;
    using E3 = Ew<C, M>;
    char* s_x = smem;
    char* s_e = smem + MR * XS;
    float* s_part = (float*)(s_e + MR * ES);      // [slots][2][16] per-wave partial group sums / sums of squares
    float* s_mr = s_part + SIR_SLOTS * 2 * SIR_G; // [IPW][16][2] mean, rstd
    float* s_p = s_mr + 2 * IPW * SIR_G;          // all per-channel parameters of the block, fetched once
    constexpr int P_BE = 0, P_G1 = E, P_B1 = 2 * E, P_BD = 3 * E, P_G2 = 4 * E, P_B2 = 5 * E, P_WD = 6 * E, P_BP = 15 * E,
                  P_G3 = 15 * E + C, P_B3 = 15 * E + 2 * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * IPW;
    const long long row0 = (long long)n0 * HW;    // first global pixel row of this workgroup
    const int rows_valid = IPW == 1 ? M : min(M, (a.N - n0) * HW);
    auto live = [&](int row) { return IPW == 1 || row < rows_valid; };      // one image per workgroup: never ragged, the row checks fold away
    const bool rec = a.raw_e != nullptr;

    SIR_STAMP(0);
    // ---- everything small the block will need, requested now so that one memory round trip covers it all
    for (int i = tid; i < E; i += SIR_THREADS) {
        s_p[P_BE + i] = a.be[i]; s_p[P_G1 + i] = a.g1[i]; s_p[P_B1 + i] = a.b1[i];
        s_p[P_BD + i] = a.bd[i]; s_p[P_G2 + i] = a.g2[i]; s_p[P_B2 + i] = a.b2[i];
    }
    for (int i = tid; i < 9 * E; i += SIR_THREADS) s_p[P_WD + i] = a.wd[i];
    for (int i = tid; i < C; i += SIR_THREADS) { s_p[P_BP + i] = a.bp[i]; s_p[P_G3 + i] = a.g3[i]; s_p[P_B3 + i] = a.b3[i]; }
    const int v1 = tid % E1::VPR, r1 = tid / E1::VPR;
    const int v3 = tid % E3::VPR, r3 = tid / E3::VPR;
    unsigned mk_e[E1::NV], mk_d[E1::NV];
#pragma unroll
    for (int k = 0; k < E1::NV; ++k) {
        const int row = r1 + k * E1::RPP;
        const long long e = (row0 + row) * E + v1 * 8;
        mk_e[k] = mk_d[k] = 0xFFu;
        if (a.m_e && live(row)) { mk_e[k] = a.m_e[e >> 3]; mk_d[k] = a.m_d[e >> 3]; }
    }
    bf16x8 rx[E3::NV];                            // clean block input for the residual
#pragma unroll
    for (int k = 0; k < E3::NV; ++k) {
        const int row = r3 + k * E3::RPP;
#pragma unroll
        for (int j = 0; j < 8; ++j) rx[k][j] = (bf16)0.f;
        if (live(row)) rx[k] = *(const bf16x8*)(a.x + (row0 + row) * C + v3 * 8);
    }
    constexpr int MT = MR / 32;
    bf16x8 bfrag1[GemmK<C>::KCH];                 // first weight chunk of this wave's first expansion tile
    if (wave < MT * (E / 32)) gemm_prefetch<C>(bfrag1, a.we + (long long)((wave / MT) * 32 + (lane & 31)) * C, lane);

    // ---- block input -> LDS (dropout of the expansion input applied here; the residual uses the clean input)
    for (int idx = tid; idx < M * (C / 8); idx += SIR_THREADS) {
        const int row = idx / (C / 8), cv = idx % (C / 8);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
        if (live(row)) {
            const long long e = (row0 + row) * C + cv * 8;
            v = *(const bf16x8*)(a.x + e);
            if (a.m_in) {
                const unsigned m = a.m_in[e >> 3];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (m >> j) & 1 ? (bf16)((float)v[j] * a.mscale) : (bf16)0.f;
            }
        }
        *(bf16x8*)(s_x + row * XS + cv * 16) = v;
    }
    __syncthreads();
    SIR_STAMP(1);

    // ---- expand: [M][C] x [E][C]^T -> raw_e (bf16) in LDS
    {
        constexpr int TT = MT * (E / 32);
#pragma unroll 1
        for (int t = wave; t < TT; t += NWV) {
            const int mt = t % MT, nt = t / MT;
            const int col = nt * 32 + (lane & 31);
            if (t != wave) gemm_prefetch<C>(bfrag1, a.we + (long long)col * C, lane);
            const f32x16 acc = gemm_tile<C, XS>(s_x, mt * 32, a.we + (long long)col * C, lane, bfrag1);
            const float b = s_p[P_BE + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_e + row * ES + col * 2) = (bf16)(acc[r] + b);
            }
        }
    }
    __syncthreads();
    SIR_STAMP(2);

    // ---- elementwise phases over the [M][E] tile ----------------------------------------------------------------
    bf16x8 xr[E1::NV];
    // statistics of a raw tile held in xr[]; optional copy of the raw values to global memory
    auto stats_phase = [&](bf16* raw_out, float* st_out) {
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) {
            const int row = r1 + k * E1::RPP;
            if (raw_out && live(row)) *(bf16x8*)(raw_out + (row0 + row) * E + v1 * 8) = xr[k];
            pair_sums(xr[k], s, ss);
            if constexpr (IPW > 1) {
                group_store<E, M>(s, ss, k * NWV + wave, v1, lane, s_part);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
            }
        }
        SIR_STAMP(10);
        if constexpr (IPW == 1) group_store<E, M>(s, ss, wave, v1, lane, s_part);
        SIR_STAMP(11);
        // the tile stays packed (bf16) across the barriers; otherwise the fp32 copies made for the sums are kept alive
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) asm volatile("" : "+v"(xr[k]));
        __syncthreads();
        SIR_STAMP(12);
        group_finish<E, M, HW, IPW>(tid, s_part, s_mr, a.eps, st_out, n0, a.N);
        SIR_STAMP(13);
        __syncthreads();
    };
    // normalise + activation + dropout of xr[] -> LDS tile (in place) and optionally global memory
    auto apply_phase = [&](const float* gamma, const float* beta, bool mask, const unsigned (&mk)[E1::NV], bf16* y_out, const float* ab) {
        const float bsc = ACT_HAS_BETA(ACT) ? *ab : 1.f;
        float ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ga[j] = gamma[v1 * 8 + j]; be[j] = beta[v1 * 8 + j]; }
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) {
            const int row = r1 + k * E1::RPP;
            const int img = IPW > 1 ? row / HW : 0;
            const long long e = (row0 + row) * E + v1 * 8;
            const unsigned m = mk[k];
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = (v1 * 8 + j) / E1::CPG;
                const float mean = s_mr[(img * SIR_G + g) * 2], rstd = s_mr[(img * SIR_G + g) * 2 + 1];
                const float sc = ga[j] * rstd, sh = be[j] - mean * sc;
                float z = act_c<ACT>((float)xr[k][j] * sc + sh, bsc);
                if (mask) z = (m >> j) & 1 ? z * a.mscale : 0.f;
                o[j] = (bf16)z;
            }
            *(bf16x8*)(s_e + row * ES + v1 * 16) = o;
            if (y_out && live(row)) *(bf16x8*)(y_out + e) = o;
        }
    };

#pragma unroll
    for (int k = 0; k < E1::NV; ++k) xr[k] = *(const bf16x8*)(s_e + (r1 + k * E1::RPP) * ES + v1 * 16);
    stats_phase(rec ? a.raw_e : nullptr, rec ? a.st1 : nullptr);
    SIR_STAMP(3);
    apply_phase(s_p + P_G1, s_p + P_B1, a.m_e != nullptr, mk_e, rec ? a.y_e : nullptr, a.ab1);
    __syncthreads();
    SIR_STAMP(4);

    // ---- depthwise 3x3 on the LDS tile -> xr[] (raw, bf16)
    {
        float acc[E1::NV][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float b = s_p[P_BD + v1 * 8 + j];
#pragma unroll
            for (int k = 0; k < E1::NV; ++k) acc[k][j] = b;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            float w[8];
            const f32x4 w0 = *(const f32x4*)(s_p + P_WD + tap * E + v1 * 8), w1 = *(const f32x4*)(s_p + P_WD + tap * E + v1 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = w0[j]; w[4 + j] = w1[j]; }
#pragma unroll
            for (int k = 0; k < E1::NV; ++k) {
                const int row = r1 + k * E1::RPP;
                const int p = row % HW, py = p / W, px = p % W;
                const int ny = py + dy, nx = px + dx;
                if (ny >= 0 && ny < W && nx >= 0 && nx < W) {
                    const bf16x8 nv = *(const bf16x8*)(s_e + (row + dy * W + dx) * ES + v1 * 16);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[k][j] += (float)nv[j] * w[j];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < E1::NV; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[k][j] = (bf16)acc[k][j];
    }
    SIR_STAMP(5);
    bf16x8 bfrag2[GemmK<E>::KCH];                 // first weight chunk of this wave's first projection tile
    if (wave < MT * (C / 32)) gemm_prefetch<E>(bfrag2, a.wp + (long long)((wave / MT) * 32 + (lane & 31)) * E, lane);
    stats_phase(rec ? a.raw_d : nullptr, rec ? a.st2 : nullptr);     // its first barrier also ends all neighbour reads
    SIR_STAMP(6);
    apply_phase(s_p + P_G2, s_p + P_B2, a.m_e != nullptr, mk_d, rec ? a.y_d : nullptr, a.ab2);
    __syncthreads();
    SIR_STAMP(7);

    // ---- project: [M][E] x [C][E]^T -> raw_o (bf16) into the LDS region of the (no longer needed) input tile
    {
        constexpr int TT = MT * (C / 32);
#pragma unroll 1
        for (int t = wave; t < TT; t += NWV) {
            const int mt = t % MT, nt = t / MT;
            const int col = nt * 32 + (lane & 31);
            if (t != wave) gemm_prefetch<E>(bfrag2, a.wp + (long long)col * E, lane);
            const f32x16 acc = gemm_tile<E, ES>(s_e, mt * 32, a.wp + (long long)col * E, lane, bfrag2);
            const float b = s_p[P_BP + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                *(bf16*)(s_x + row * XS + col * 2) = (bf16)(acc[r] + b);
            }
        }
    }
    __syncthreads();
    SIR_STAMP(8);

    // ---- last GroupNorm (no activation) + residual -> block output
    {
        bf16x8 xo[E3::NV];
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const int row = r3 + k * E3::RPP;
            xo[k] = *(const bf16x8*)(s_x + row * XS + v3 * 16);
            const long long e = (row0 + row) * C + v3 * 8;
            if (rec && live(row)) *(bf16x8*)(a.raw_o + e) = xo[k];
            pair_sums(xo[k], s, ss);
            if constexpr (IPW > 1) {
                group_store<C, M>(s, ss, k * NWV + wave, v3, lane, s_part);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
            }
        }
        if constexpr (IPW == 1) group_store<C, M>(s, ss, wave, v3, lane, s_part);
        __syncthreads();
        group_finish<C, M, HW, IPW>(tid, s_part, s_mr, a.eps, rec ? a.st3 : nullptr, n0, a.N);
        __syncthreads();
        float ga[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ga[j] = s_p[P_G3 + v3 * 8 + j]; be[j] = s_p[P_B3 + v3 * 8 + j]; }
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const int row = r3 + k * E3::RPP;
            if (!live(row)) continue;
            const int img = IPW > 1 ? row / HW : 0;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int g = (v3 * 8 + j) / E3::CPG;
                const float mean = s_mr[(img * SIR_G + g) * 2], rstd = s_mr[(img * SIR_G + g) * 2 + 1];
                const float sc = ga[j] * rstd, sh = be[j] - mean * sc;
                o[j] = (bf16)((float)xo[k][j] * sc + sh + (float)rx[k][j]);
            }
            *(bf16x8*)(a.y + (row0 + row) * C + v3 * 8) = o;
        }
    }
    SIR_STAMP(9);
}

// Waves per SIMD the register allocation must leave room for.  The 16x16 shape, whose natural allocation is 137
// registers, is held to 128 (24 bytes of spills outside any loop) so that two workgroups share a CU: with one workgroup
// per CU the latencies of its serial phases are exposed, and at sampling batch sizes there are several workgroups per CU
// to overlap (4883 against 4688 sampled images/s).  The 8x8 / 4x4 shapes of wider networks spill more and showed no gain.
constexpr int sir_fwd_waves(int c, int w) { return (c == 32 && w == 16) ? 4 : 2; }

// LEAN = inference (nothing is saved for a backward pass, no dropout masks; the host checks): with the pointers known to be
// null at compile time, the addresses, mask words and stores of the recording path leave the kernel (and its registers).
template <bool LEAN>
__device__ __forceinline__ SirArgs sir_lean_args(SirArgs a) {
    if constexpr (LEAN) {
        a.m_in = nullptr; a.m_e = nullptr; a.m_d = nullptr;
        a.raw_e = nullptr; a.y_e = nullptr; a.st1 = nullptr; a.raw_d = nullptr; a.y_d = nullptr; a.st2 = nullptr; a.raw_o = nullptr; a.st3 = nullptr;
    }
    return a;
}

template <int C, int W, bool LEAN>
__global__ __launch_bounds__(SIR_THREADS, sir_fwd_waves(C, W)) void sir_fwd_kernel(const SirArgs a_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SirArgs a = sir_lean_args<LEAN>(a_);
    switch (a.act) {
        case ACT_GELU: sir_fwd_body<C, W, ACT_GELU>(a, smem); break;
        case ACT_SILU: sir_fwd_body<C, W, ACT_SILU>(a, smem); break;
        case ACT_SWISH: sir_fwd_body<C, W, ACT_SWISH>(a, smem); break;
        case ACT_HARDSWISH: sir_fwd_body<C, W, ACT_HARDSWISH>(a, smem); break;
        default: sir_fwd_body<C, W, ACT_RELU>(a, smem); break;
    }
}


// Variant for images too large to keep both the block input and the expanded tensor in LDS (32x32 pixels at C = 32:
// the expanded [1024][64] tile alone takes 144 KB).  One workgroup still owns one image; the expansion GEMM takes its
// A fragments straight from global memory (each input element is used by exactly one fragment load), the projection
// keeps all of a wave's output tiles in accumulators until every wave has finished reading the expanded tile, and then
// reuses that LDS for the raw projection output.  Elementwise phases walk 16 vectors per thread.
template <int C, int W, int ACT>
__device__ __forceinline__ void sir_fwd_big_body(const SirArgs& a, char* smem) {
    constexpr int E = 2 * C, HW = W * W, M = HW;
    constexpr int XS = C * 2 + 16, ES = E * 2 + 16;
    constexpr int NWV = SIR_THREADS / 64, MT = M / 32, TPW = MT / NWV;
    constexpr int NTC = C / 32;                    // output-channel tiles of the projection
    static_assert(MT % NWV == 0 && C % 32 == 0 && TPW * NTC <= 4, "big variant: M tiles dealt evenly, at most four accumulator tiles per wave");
    using E1 = Ew<E, M>;
    using E3 = Ew<C, M>;
    char* s_e = smem;                              // [M][ES]; later [M][XS] raw projection output
    float* s_part = (float*)(smem + M * ES);       // [waves][2][16]
    float* s_mr = s_part + NWV * 2 * SIR_G;        // [16][2]
    float* s_p = s_mr + 2 * SIR_G;
    constexpr int P_BE = 0, P_G1 = E, P_B1 = 2 * E, P_BD = 3 * E, P_G2 = 4 * E, P_B2 = 5 * E, P_WD = 6 * E, P_BP = 15 * E,
                  P_G3 = 15 * E + C, P_B3 = 15 * E + 2 * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long n0 = blockIdx.x;
    const long long row0 = n0 * HW;
    const bool rec = a.raw_e != nullptr;
    // workgroup-uniform bases of this image's rows; lanes add 32-bit byte offsets (ld_at / st_at)
    const bf16* const xb = a.x + row0 * C;
    bf16* const yb = a.y + row0 * C;
    bf16* const raw_eb = a.raw_e + row0 * E; bf16* const y_eb = a.y_e + row0 * E;
    bf16* const raw_db = a.raw_d + row0 * E; bf16* const y_db = a.y_d + row0 * E;
    bf16* const raw_ob = a.raw_o + row0 * C;
    const unsigned char* const m_inb = a.m_in + (row0 * C >> 3);
    const unsigned char* const m_eb = a.m_e + (row0 * E >> 3);
    const unsigned char* const m_db = a.m_d + (row0 * E >> 3);

    for (int i = tid; i < E; i += SIR_THREADS) {
        s_p[P_BE + i] = a.be[i]; s_p[P_G1 + i] = a.g1[i]; s_p[P_B1 + i] = a.b1[i];
        s_p[P_BD + i] = a.bd[i]; s_p[P_G2 + i] = a.g2[i]; s_p[P_B2 + i] = a.b2[i];
    }
    for (int i = tid; i < 9 * E; i += SIR_THREADS) s_p[P_WD + i] = a.wd[i];
    for (int i = tid; i < C; i += SIR_THREADS) { s_p[P_BP + i] = a.bp[i]; s_p[P_G3 + i] = a.g3[i]; s_p[P_B3 + i] = a.b3[i]; }

    // ---- expand: A fragments from global memory (dropout of the block input applied to the fragment)
    {
        constexpr int KS = C / 16;
        bf16x8 af[TPW][KS];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned e = (unsigned)(((wave + NWV * i) * 32 + (lane & 31)) * C + ks * 16 + (lane >> 5) * 8);
                bf16x8 v = ld_at<bf16x8>(xb, e * 2);
                if (a.m_in) {
                    const unsigned m = ld_at<unsigned char>(m_inb, e >> 3);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (m >> j) & 1 ? (bf16)((float)v[j] * a.mscale) : (bf16)0.f;
                }
                af[i][ks] = v;
            }
        __syncthreads();
#pragma unroll 1
        for (int nt = 0; nt < E / 32; ++nt) {
            const int col = nt * 32 + (lane & 31);
            bf16x8 bf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[ks] = *(const bf16x8*)(a.we + (long long)col * C + ks * 16 + (lane >> 5) * 8);
            const float b = s_p[P_BE + col];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ks], bf[ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave + NWV * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    *(bf16*)(s_e + row * ES + col * 2) = (bf16)(acc[r] + b);
                }
            }
        }
    }
    __syncthreads();

    const int v1 = tid % E1::VPR, r1 = tid / E1::VPR;
    bf16x8 xr[E1::NV];
    auto stats_phase = [&](bf16* raw_out, float* st_out) {
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) {
            const int row = r1 + k * E1::RPP;
            if (raw_out) st_at(raw_out, (unsigned)(row * E + v1 * 8) * 2, xr[k]);
            pair_sums(xr[k], s, ss);
        }
        group_store<E, M>(s, ss, wave, v1, lane, s_part);
        // the tile stays packed (bf16) across the barriers: without this the fp32 copies made for the sums above are kept
        // for the normalisation below, 128 registers of them
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) asm volatile("" : "+v"(xr[k]));
        __syncthreads();
        group_finish<E, M, HW, 1>(tid, s_part, s_mr, a.eps, st_out, n0, a.N);
        __syncthreads();
    };
    auto apply_phase = [&](const float* gamma, const float* beta, const unsigned char* mask, bf16* y_out, const float* ab) {
        const float bsc = ACT_HAS_BETA(ACT) ? *ab : 1.f;
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (v1 * 8 + j) / E1::CPG;
            const float mean = s_mr[g * 2], rstd = s_mr[g * 2 + 1];
            sc[j] = gamma[v1 * 8 + j] * rstd; sh[j] = beta[v1 * 8 + j] - mean * sc[j];
        }
        unsigned mk[E1::NV];
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) mk[k] = mask ? ld_at<unsigned char>(mask, (unsigned)((r1 + k * E1::RPP) * E + v1 * 8) >> 3) : 0xFFu;
#pragma unroll
        for (int k = 0; k < E1::NV; ++k) {
            const int row = r1 + k * E1::RPP;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float z = act_c<ACT>((float)xr[k][j] * sc[j] + sh[j], bsc);
                if (mask) z = (mk[k] >> j) & 1 ? z * a.mscale : 0.f;
                o[j] = (bf16)z;
            }
            asm volatile("" : "+v"(o));                    // one vector after the other: interleaved activation chains spill
            *(bf16x8*)(s_e + row * ES + v1 * 16) = o;
            if (y_out) st_at(y_out, (unsigned)(row * E + v1 * 8) * 2, o);
        }
    };

#pragma unroll
    for (int k = 0; k < E1::NV; ++k) xr[k] = *(const bf16x8*)(s_e + (r1 + k * E1::RPP) * ES + v1 * 16);
    stats_phase(rec ? raw_eb : nullptr, rec ? a.st1 : nullptr);
    apply_phase(s_p + P_G1, s_p + P_B1, a.m_e ? m_eb : nullptr, rec ? y_eb : nullptr, a.ab1);
    __syncthreads();

    // ---- depthwise 3x3 from the LDS tile, four vectors at a time (accumulator registers)
    {
        constexpr int KB = 4;
        float bd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bd[j] = s_p[P_BD + v1 * 8 + j];
#pragma unroll
        for (int k0 = 0; k0 < E1::NV; k0 += KB) {
            float acc[KB][8];
#pragma unroll
            for (int k = 0; k < KB; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[k][j] = bd[j];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                float w[8];
                const f32x4 w0 = *(const f32x4*)(s_p + P_WD + tap * E + v1 * 8), w1 = *(const f32x4*)(s_p + P_WD + tap * E + v1 * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { w[j] = w0[j]; w[4 + j] = w1[j]; }
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const int row = r1 + (k0 + k) * E1::RPP;
                    const int py = row / W, px = row % W;
                    const int ny = py + dy, nx = px + dx;
                    if (ny >= 0 && ny < W && nx >= 0 && nx < W) {
                        const bf16x8 nv = *(const bf16x8*)(s_e + (row + dy * W + dx) * ES + v1 * 16);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[k][j] += (float)nv[j] * w[j];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[k][j];
                xr[k0 + k] = o;
                asm volatile("" : "+v"(xr[k0 + k]));       // convert now: keeps the chunks from being interleaved (which spills)
            }
        }
    }
    bf16x8 bf2[E / 16];                            // projection weights of this lane's output channel in the first N tile
#pragma unroll
    for (int ks = 0; ks < E / 16; ++ks) bf2[ks] = *(const bf16x8*)(a.wp + (long long)(lane & 31) * E + ks * 16 + (lane >> 5) * 8);
    stats_phase(rec ? raw_db : nullptr, rec ? a.st2 : nullptr);     // its first barrier also ends all neighbour reads
    apply_phase(s_p + P_G2, s_p + P_B2, a.m_e ? m_db : nullptr, rec ? y_db : nullptr, a.ab2);
    __syncthreads();

    // ---- project: all of this wave's tiles stay in accumulators until every wave has read the expanded tile
    {
        f32x16 acc[TPW][NTC];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            if (nt > 0) {                          // wider blocks: the next N tile's weights (the A fragments come from LDS again)
#pragma unroll
                for (int ks = 0; ks < E / 16; ++ks)
                    bf2[ks] = *(const bf16x8*)(a.wp + (long long)(nt * 32 + (lane & 31)) * E + ks * 16 + (lane >> 5) * 8);
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][nt][r] = 0.f;
                const char* ap = s_e + ((wave + NWV * i) * 32 + (lane & 31)) * ES + (lane >> 5) * 16;
#pragma unroll
                for (int ks = 0; ks < E / 16; ++ks) {
                    const bf16x8 av = *(const bf16x8*)(ap + ks * 32);
                    acc[i][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bf2[ks], acc[i][nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const float b = s_p[P_BP + nt * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wave + NWV * i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    *(bf16*)(s_e + row * XS + (nt * 32 + (lane & 31)) * 2) = (bf16)(acc[i][nt][r] + b);
                }
        }
    }
    __syncthreads();

    // ---- last GroupNorm (no activation) + residual
    {
        const int v3 = tid % E3::VPR, r3 = tid / E3::VPR;
        bf16x8 xo[E3::NV];
        float s[8], ss[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const int row = r3 + k * E3::RPP;
            xo[k] = *(const bf16x8*)(s_e + row * XS + v3 * 16);
            if (rec) st_at(raw_ob, (unsigned)(row * C + v3 * 8) * 2, xo[k]);
            pair_sums(xo[k], s, ss);
        }
        group_store<C, M>(s, ss, wave, v3, lane, s_part);
        bf16x8 rx[E3::NV];
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) rx[k] = ld_at<bf16x8>(xb, (unsigned)((r3 + k * E3::RPP) * C + v3 * 8) * 2);
        __syncthreads();
        group_finish<C, M, HW, 1>(tid, s_part, s_mr, a.eps, rec ? a.st3 : nullptr, n0, a.N);
        __syncthreads();
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (v3 * 8 + j) / E3::CPG;
            const float mean = s_mr[g * 2], rstd = s_mr[g * 2 + 1];
            sc[j] = s_p[P_G3 + v3 * 8 + j] * rstd; sh[j] = s_p[P_B3 + v3 * 8 + j] - mean * sc[j];
        }
#pragma unroll
        for (int k = 0; k < E3::NV; ++k) {
            const int row = r3 + k * E3::RPP;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)xo[k][j] * sc[j] + sh[j] + (float)rx[k][j]);
            st_at(yb, (unsigned)(row * C + v3 * 8) * 2, o);
        }
    }
}

template <int C, int W, bool LEAN>
__global__ __launch_bounds__(SIR_THREADS) void sir_fwd_big_kernel(const SirArgs a_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SirArgs a = sir_lean_args<LEAN>(a_);
    switch (a.act) {
        case ACT_GELU: sir_fwd_big_body<C, W, ACT_GELU>(a, smem); break;
        case ACT_SILU: sir_fwd_big_body<C, W, ACT_SILU>(a, smem); break;
        case ACT_SWISH: sir_fwd_big_body<C, W, ACT_SWISH>(a, smem); break;
        case ACT_HARDSWISH: sir_fwd_big_body<C, W, ACT_HARDSWISH>(a, smem); break;
        default: sir_fwd_big_body<C, W, ACT_RELU>(a, smem); break;
    }
}

template <int C, int W, bool LEAN>
static int sir_big_launch_as(const SirArgs& a, hipStream_t stream) {
    constexpr int E = 2 * C, M = W * W;
    constexpr int lds = M * (E * 2 + 16) + ((SIR_THREADS / 64) * 2 * SIR_G + 2 * SIR_G + 15 * E + 3 * C) * 4;
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    auto k = sir_fwd_big_kernel<C, W, LEAN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            ddpm_set_error("sir_forward: cannot reserve %d bytes of LDS", lds);
            return DDPM_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(a.N), dim3(SIR_THREADS), lds, stream, a);
    return ddpm_check_launch("sir_forward");
}

static bool sir_is_lean(const SirArgs& a) { return !a.raw_e && !a.m_in && !a.m_e && !a.m_d; }

template <int C, int W>
static int sir_big_launch(const SirArgs& a, hipStream_t stream) {
    return sir_is_lean(a) ? sir_big_launch_as<C, W, true>(a, stream) : sir_big_launch_as<C, W, false>(a, stream);
}

template <int C, int W, bool LEAN>
static int sir_launch_as(const SirArgs& a, hipStream_t stream) {
    constexpr int E = 2 * C, HW = W * W, IPW = sir_ipw(HW), M = sir_tile_rows(IPW * HW);
    constexpr int lds = M * (C * 2 + 16) + M * (E * 2 + 16) + (SIR_SLOTS * 2 * SIR_G + 2 * IPW * SIR_G + 15 * E + 3 * C) * 4;
    static_assert(IPW == 1 || (Ew<E, M>::NV <= 4 && Ew<C, M>::NV <= 4), "more passes than partial-sum slots");
    static_assert(lds <= 160 * 1024, "tile does not fit LDS");
    auto k = sir_fwd_kernel<C, W, LEAN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            ddpm_set_error("sir_forward: cannot reserve %d bytes of LDS", lds);
            return DDPM_EHIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3((a.N + IPW - 1) / IPW), dim3(SIR_THREADS), lds, stream, a);
    return ddpm_check_launch("sir_forward");
}

template <int C, int W>
static int sir_launch(const SirArgs& a, hipStream_t stream) {
    return sir_is_lean(a) ? sir_launch_as<C, W, true>(a, stream) : sir_launch_as<C, W, false>(a, stream);
}

extern "C" int ddpm_sir_supported(int C, int H, int W, int E, int groups, int act) {
    if (H != W || E != 2 * C || groups != SIR_G) return 0;
    if (act != ACT_GELU && act != ACT_SILU && act != ACT_RELU && !ACT_HAS_BETA(act)) return 0;
    return (C == 32 && W == 16) || (C == 64 && W == 8) || (C == 128 && W == 4) || (C == 256 && W == 2) ||
           (C == 128 && W == 8) || (C == 256 && W == 4) || (C == 32 && W == 32) || (C == 64 && W == 16) || (C == 256 && W == 8) ||
           (C == 128 && W == 16);
}

// 1 when the backward kernel (ddpm_sir_backward) also covers the shape
extern "C" int ddpm_sir_backward_supported(int C, int H, int W, int E, int groups, int act) {
    if (ACT_HAS_BETA(act) && !sir_bwd_takes_scalar_acts(C, W)) return 0;
    if ((C == 256 && W == 8) || (C == 128 && W == 16)) return 0;    // the two deepest stages of 128x128 images: forward only
    return ddpm_sir_supported(C, H, W, E, groups, act);
}

extern "C" int ddpm_sir_forward(const ddpm_sir_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(d && d->x && d->y && d->we && d->be && d->g1 && d->b1 && d->wd && d->bd && d->g2 && d->b2 && d->wp && d->bp &&
                     d->g3 && d->b3, "sir_forward: null pointer");
    DDPM_REQUIRE(d->N > 0, "sir_forward: N must be positive");
    DDPM_REQUIRE(ddpm_sir_supported(d->C, d->H, d->W, d->E, d->groups, d->act),
                 "sir_forward: unsupported shape C=%d %dx%d E=%d groups=%d act=%d (use the unfused kernels)", d->C, d->H, d->W, d->E,
                 d->groups, d->act);
    DDPM_REQUIRE(!ACT_HAS_BETA(d->act) || (d->act_b1 && d->act_b2), "sir_forward: swish / hardswish need the two activation modules' scalars (act_b1, act_b2)");
    const bool rec = d->raw_e != nullptr;
    if (rec) DDPM_REQUIRE(d->y_e && d->st1 && d->raw_d && d->y_d && d->st2 && d->raw_o && d->st3, "sir_forward: incomplete set of saved-tensor pointers");
    SirArgs a;
    a.x = (const bf16*)d->x; a.y = (bf16*)d->y;
    a.we = (const bf16*)d->we; a.be = d->be; a.g1 = d->g1; a.b1 = d->b1;
    a.wd = d->wd; a.bd = d->bd; a.g2 = d->g2; a.b2 = d->b2;
    a.wp = (const bf16*)d->wp; a.bp = d->bp; a.g3 = d->g3; a.b3 = d->b3;
    a.m_in = d->m_in; a.m_e = d->m_e; a.m_d = d->m_d; a.mscale = d->mask_scale;
    a.raw_e = (bf16*)d->raw_e; a.y_e = (bf16*)d->y_e; a.st1 = d->st1;
    a.raw_d = (bf16*)d->raw_d; a.y_d = (bf16*)d->y_d; a.st2 = d->st2;
    a.raw_o = (bf16*)d->raw_o; a.st3 = d->st3;
    a.N = d->N; a.act = d->act; a.eps = d->eps;
    a.ab1 = d->act_b1; a.ab2 = d->act_b2;
    if (d->C == 32 && d->W == 32) return sir_big_launch<32, 32>(a, stream);
    if (d->C == 128 && d->W == 16) return sir_big_launch<128, 16>(a, stream);
    if (d->C == 32 && d->W == 16) return sir_launch<32, 16>(a, stream);
    if (d->C == 64 && d->W == 8) return sir_launch<64, 8>(a, stream);
    if (d->C == 128 && d->W == 4) return sir_launch<128, 4>(a, stream);
    if (d->C == 256 && d->W == 2) return sir_launch<256, 2>(a, stream);
    if (d->C == 128 && d->W == 8) return sir_launch<128, 8>(a, stream);
    if (d->C == 64 && d->W == 16) return sir_launch<64, 16>(a, stream);
    if (d->C == 256 && d->W == 8) return sir_launch<256, 8>(a, stream);
    return sir_launch<256, 4>(a, stream);
}
