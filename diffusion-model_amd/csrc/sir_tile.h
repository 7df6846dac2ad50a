// Tile bookkeeping and the small GEMM shared by the fused inverted-residual kernels (forward and backward).
#pragma once
#include "common.h"

constexpr int SIR_G = 16;            // GroupNorm groups (the reference's base_groups_norm * 4 default)
constexpr int SIR_THREADS = 512;
constexpr int SIR_SLOTS = 4 * (SIR_THREADS / 64);   // partial-sum slots: up to 4 passes x 8 waves

// Images per workgroup: one for images of >= 32 pixels; small images are batched so that a tile has 32 rows (one MFMA
// row block), except 2x2 images: four of them (16 rows) - at 256+ channels the elementwise phases, not the half-empty
// MFMA tile, are what a workgroup spends its time on, and twice as many workgroups share them.
constexpr int sir_ipw(int hw) { return hw >= 32 ? 1 : (hw == 4 ? 4 : 32 / hw); }
// Rows the LDS tiles are allocated (and the MFMA tiles computed) for: at least one 32-row block; rows >= M are never read back.
constexpr int sir_tile_rows(int m) { return m < 32 ? 32 : m; }

// Shapes whose one-launch backward also takes swish / hardswish (it then sums the scalar's gradient as well): all but the two
// widest tiles, which have no registers to spare for it.
constexpr bool sir_bwd_takes_scalar_acts(int c, int w) { return !((c == 64 && w == 16) || (c == 256 && w == 4)); }

template <int ACT>
__device__ __forceinline__ float act_c(float z) {
    if constexpr (ACT == ACT_GELU) return gelu_fwd(z);
    else if constexpr (ACT == ACT_SILU) return z * sigmoid_fast(z);
    else if constexpr (ACT == ACT_RELU) return z > 0.f ? z : 0.f;
    else return z;
}
// the same with the learnable scalar b of swish / hardswish (neural_nets/activations.py:25-63 of the reference); the other
// codes ignore it
template <int ACT>
__device__ __forceinline__ float act_c(float z, float b) {
    if constexpr (ACT == ACT_SWISH) return z * sigmoid_fast(b * z);
    else if constexpr (ACT == ACT_HARDSWISH) return z * hardsigmoid_of(b * z);
    else return act_c<ACT>(z);
}

// Shape bookkeeping of one elementwise phase over an [M][CH] bf16 tile: thread -> (row, 8-channel vector).
template <int CH, int M>
struct Ew {
    static constexpr int VPR = CH / 8;                       // vectors per row
    static constexpr int RPP = SIR_THREADS / VPR;            // rows per pass
    static constexpr int NV = M * VPR / SIR_THREADS;         // vectors per thread
    static constexpr int CPG = CH / SIR_G;                   // channels per group
    static constexpr int GPV = CPG >= 8 ? 1 : 8 / CPG;       // groups inside one vector
    static constexpr int VPG = CPG >= 8 ? CPG / 8 : 1;       // vectors spanning one group
    typedef bf16x8 Vecs[NV];                                 // the vectors one thread holds
    static_assert(NV >= 1 && VPR <= 64 && (SIR_THREADS % VPR) == 0, "unsupported tile shape");
};

// One 32x32 output tile of A[M][K] (LDS, row stride AS bytes) x B[N][K]^T (global bf16, row stride K elements).
// The weight fragments of the first K chunk are loaded by the caller (gemm_prefetch) ahead of the barrier that
// publishes A, so their L2 latency is not on the critical path.
template <int K>
struct GemmK {
    static constexpr int KS = K / 16, KCH = KS < 8 ? KS : 8;
};

template <int K>
__device__ __forceinline__ void gemm_prefetch(bf16x8 (&bv)[GemmK<K>::KCH], const bf16* wrow, int lane) {
    const bf16* bp = wrow + (lane >> 5) * 8;
#pragma unroll
    for (int u = 0; u < GemmK<K>::KCH; ++u) bv[u] = *(const bf16x8*)(bp + u * 16);
}

template <int K, int AS>
__device__ __forceinline__ f32x16 gemm_tile(const char* s_a, int m0, const bf16* wrow /* row n0 + (lane & 31) */, int lane,
                                            bf16x8 (&bv)[GemmK<K>::KCH]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int KS = GemmK<K>::KS, KCH = GemmK<K>::KCH;
    const char* ap = s_a + (m0 + (lane & 31)) * AS + (lane >> 5) * 16;
    const bf16* bp = wrow + (lane >> 5) * 8;
#pragma unroll 1
    for (int k0 = 0; k0 < KS; k0 += KCH) {
        if (k0 > 0) {
#pragma unroll
            for (int u = 0; u < KCH; ++u) bv[u] = *(const bf16x8*)(bp + (k0 + u) * 16);
        }
#pragma unroll
        for (int u = 0; u < KCH; ++u) {
            const bf16x8 av = *(const bf16x8*)(ap + (k0 + u) * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[u], acc, 0, 0, 0);
        }
    }
    return acc;
}

