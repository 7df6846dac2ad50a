// Multi-head attention core (softmax(Q K^T / sqrt(dh)) V) forward and backward on bf16 MFMA, and the per-sample
// normalisation of the reference's ConditionalModule (constituent_blocks.py:402-437: nn.MultiheadAttention followed by
// GroupNorm(1, C) of a residual sum, then Linear -> activation -> GroupNorm(1, C) of a second residual sum).
//
// Attention.  Every matrix product is arranged so that the OWNER of a softmax row is an MFMA output COLUMN:
//   forward / dQ pass : S^T = K Q^T  -> lane (l % 32) owns one query, its 16 accumulators are 16 keys of the tile
//   dK,dV pass        : S   = Q K^T  -> lane (l % 32) owns one key, its accumulators are 16 queries
// With v_mfma_f32_32x32x16_bf16 an accumulator register i of lane l is row (i/4)*8 + (l/32)*4 + i%4, and a B operand wants
// rows (l/32)*8 .. +7 of the same column: the contraction index may be permuted freely as long as both operands use the
// same permutation, so the probabilities go from accumulators to the next product's B operand without leaving registers;
// the A operand (V^T, K^T, Q^T, dO^T: head-dim x tokens) is read from a transposed copy in LDS with the matching
// permutation.  The running max / sum of the online softmax are per-lane scalars (one exchange with lane ^ 32).
// Head dims of 4 and of multiples of 8 up to 64 are handled by zero padding to DHP = 16, 32 or 64.
//
// Dropout of the attention probabilities (nn.MultiheadAttention(dropout=drop_p), constituent_blocks.py:358-366) draws one
// Philox block per 4 queries x 4 keys, one byte per element, so that both tile orientations regenerate identical masks;
// the keep probability is therefore a multiple of 1/256 and the scale is its exact reciprocal.
#include "common.h"
#include "ddpm_hip.h"

struct AttnArgs {
    const bf16 *q, *k, *v;
    int ldq, ldk, ldv;
    bf16* o;
    int ldo;
    float* lse;                // [N][heads][S]: log2 of the softmax denominator plus the row maximum, in the base-2 domain
    const bf16* dout;
    int ldd;
    bf16 *dq, *dk, *dv;
    int lddq, lddk, lddv;
    int N, S, Sk, heads, dh;
    float scale, scale2;       // 1/sqrt(dh) and that times log2(e)
    int drop_thresh;           // keep an element when its byte < drop_thresh; 256 = no dropout
    float drop_scale;
    const uint64_t* seed;
    uint32_t stream_id;
};

// 8 consecutive head-dim elements d0 .. d0+7 of one token row (zeros beyond dh, or when the row does not exist)
__device__ __forceinline__ bf16x8 load_frag(const bf16* row, int d0, int dh, bool valid) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)0.f;
    if (valid) {
        if (d0 + 8 <= dh) {
            r = *(const bf16x8*)(row + d0);
        } else if (d0 + 4 <= dh) {
            const bf16x4 t = *(const bf16x4*)(row + d0);
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = t[j];
        }
    }
    return r;
}

// LDS dst[d][tok] (row stride `stride`) = src[tok][d] for tok < SP (zeros for tok >= S), d < DHP (zeros for d >= dh)
template <int DHP>
__device__ __forceinline__ void fill_transposed(bf16* dst, int stride, const bf16* src, int ld, int S, int SP, int dh, int tid) {
    constexpr int DC = DHP / 8;
    for (int idx = tid; idx < SP * DC; idx += 256) {
        const int tok = idx / DC, dc = idx - tok * DC;
        const bool ok = tok < S;
        const bf16x8 f = load_frag(src + (long long)(ok ? tok : 0) * ld, dc * 8, dh, ok);
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[(dc * 8 + j) * stride + tok] = f[j];
    }
}

// A operand from a transposed LDS copy: row d, contraction slots (half h, j) <-> tokens tok0 + 8*(j/4) + 4*h + j%4
__device__ __forceinline__ bf16x8 lds_frag(const bf16* t, int stride, int d, int tok0, int h, bool valid) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16)0.f;
    if (valid) {
        const bf16* p = t + d * stride + tok0 + 4 * h;
        const bf16x4 lo = *(const bf16x4*)p, hi = *(const bf16x4*)(p + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    }
    return r;
}

// keep bytes of the 4x4 block (queries 4*qb .. +3) x (keys 4*kb .. +3): word = query % 4, byte = key % 4
__device__ __forceinline__ Philox4 attn_mask_block(const AttnArgs& a, uint64_t seed, int nh, int qb, int kb) {
    return philox4x32_10((uint32_t)kb, (uint32_t)qb, (uint32_t)nh, a.stream_id ^ 0x5bd1e995u, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t word_of(const Philox4& r, int i) { return i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w; }

#define MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)

template <int DHP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* vt = (bf16*)smem;
    constexpr int KC = DHP / 16, MT = (DHP + 31) / 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, n = blockIdx.z, hd = head * a.dh;
    const int SKP = (a.Sk + 31) & ~31, stride = SKP + 4;
    fill_transposed<DHP>(vt, stride, a.v + (long long)n * a.Sk * a.ldv + hd, a.ldv, a.Sk, SKP, a.dh, tid);
    __syncthreads();
    const int q = blockIdx.x * 128 + wave * 32 + col;
    const bool qv = q < a.S;
    const bf16* qrow = a.q + ((long long)n * a.S + (qv ? q : 0)) * a.ldq + hd;
    bf16x8 bq[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) bq[kc] = load_frag(qrow, kc * 16 + h * 8, a.dh, qv);
    float m = -INFINITY, lsum = 0.f;
    f32x16 oacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[mt][i] = 0.f;
    const bf16* kbase = a.k + (long long)n * a.Sk * a.ldk + hd;
    const bool drop = a.drop_thresh < 256;
    const uint64_t seed = drop ? *a.seed : 0;
    for (int kt = 0; kt < SKP; kt += 32) {
        const int key = kt + col;
        const bool kv = key < a.Sk;
        const bf16* krow = kbase + (long long)(kv ? key : 0) * a.ldk;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) acc = MFMA(load_frag(krow, kc * 16 + h * 8, a.dh, kv), bq[kc], acc);
        float s[16], mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kt + (i >> 2) * 8 + h * 4 + (i & 3);
            s[i] = kk < a.Sk ? acc[i] * a.scale2 : -INFINITY;
            mx = fmaxf(mx, s[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float rs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mn); rs += s[i]; }
        rs += __shfl_xor(rs, 32, 64);
        lsum = lsum * alpha + rs;
        m = mn;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[mt][i] *= alpha;
        if (drop) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const Philox4 r = attn_mask_block(a, seed, n * a.heads + head, q >> 2, (kt + g * 8 + h * 4) >> 2);
                const uint32_t w = word_of(r, q & 3);
#pragma unroll
                for (int j = 0; j < 4; ++j) s[g * 4 + j] = (int)((w >> (8 * j)) & 0xFFu) < a.drop_thresh ? s[g * 4 + j] * a.drop_scale : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (bf16)s[8 * c + j];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                oacc[mt] = MFMA(lds_frag(vt, stride, mt * 32 + col, kt + 16 * c, h, mt * 32 + col < DHP), pb, oacc[mt]);
        }
    }
    if (qv) {
        const float inv = 1.0f / lsum;
        bf16* orow = a.o + ((long long)n * a.S + q) * a.ldo + hd;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = mt * 32 + g * 8 + h * 4;
                if (d < a.dh) {
                    bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (bf16)(oacc[mt][g * 4 + j] * inv);
                    *(bf16x4*)(orow + d) = w;
                }
            }
        if (a.lse && h == 0) a.lse[((long long)n * a.heads + head) * a.S + q] = m + __builtin_amdgcn_logf(lsum);
    }
}

// dQ: same orientation as the forward pass (lane owns a query, loop over key tiles); K^T comes from LDS.
template <int DHP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* ktr = (bf16*)smem;
    constexpr int KC = DHP / 16, MT = (DHP + 31) / 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, n = blockIdx.z, hd = head * a.dh;
    const int SKP = (a.Sk + 31) & ~31, stride = SKP + 4;
    const bf16* kbase = a.k + (long long)n * a.Sk * a.ldk + hd;
    const bf16* vbase = a.v + (long long)n * a.Sk * a.ldv + hd;
    fill_transposed<DHP>(ktr, stride, kbase, a.ldk, a.Sk, SKP, a.dh, tid);
    __syncthreads();
    const int q = blockIdx.x * 128 + wave * 32 + col;
    const bool qv = q < a.S;
    const long long qi = (long long)n * a.S + (qv ? q : 0);
    bf16x8 bq[KC], bdo[KC];
    float dsum = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        bq[kc] = load_frag(a.q + qi * a.ldq + hd, kc * 16 + h * 8, a.dh, qv);
        bdo[kc] = load_frag(a.dout + qi * a.ldd + hd, kc * 16 + h * 8, a.dh, qv);
        const bf16x8 of = load_frag(a.o + qi * a.ldo + hd, kc * 16 + h * 8, a.dh, qv);
#pragma unroll
        for (int j = 0; j < 8; ++j) dsum += (float)bdo[kc][j] * (float)of[j];
    }
    dsum += __shfl_xor(dsum, 32, 64);
    const float lse = qv ? a.lse[((long long)n * a.heads + head) * a.S + q] : 0.f;
    f32x16 dqacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dqacc[mt][i] = 0.f;
    const bool drop = a.drop_thresh < 256;
    const uint64_t seed = drop ? *a.seed : 0;
    for (int kt = 0; kt < SKP; kt += 32) {
        const int key = kt + col;
        const bool kv = key < a.Sk;
        const bf16* krow = kbase + (long long)(kv ? key : 0) * a.ldk;
        const bf16* vrow = vbase + (long long)(kv ? key : 0) * a.ldv;
        f32x16 acc, adp;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = adp[i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            acc = MFMA(load_frag(krow, kc * 16 + h * 8, a.dh, kv), bq[kc], acc);
            adp = MFMA(load_frag(vrow, kc * 16 + h * 8, a.dh, kv), bdo[kc], adp);
        }
        float ds[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kt + (i >> 2) * 8 + h * 4 + (i & 3);
            ds[i] = kk < a.Sk ? __builtin_amdgcn_exp2f(acc[i] * a.scale2 - lse) : 0.f;     // the probability
        }
        if (drop) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const Philox4 r = attn_mask_block(a, seed, n * a.heads + head, q >> 2, (kt + g * 8 + h * 4) >> 2);
                const uint32_t w = word_of(r, q & 3);
#pragma unroll
                for (int j = 0; j < 4; ++j) adp[g * 4 + j] = (int)((w >> (8 * j)) & 0xFFu) < a.drop_thresh ? adp[g * 4 + j] * a.drop_scale : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) ds[i] *= adp[i] - dsum;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 db;
#pragma unroll
            for (int j = 0; j < 8; ++j) db[j] = (bf16)ds[8 * c + j];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                dqacc[mt] = MFMA(lds_frag(ktr, stride, mt * 32 + col, kt + 16 * c, h, mt * 32 + col < DHP), db, dqacc[mt]);
        }
    }
    if (qv) {
        bf16* row = a.dq + ((long long)n * a.S + q) * a.lddq + hd;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = mt * 32 + g * 8 + h * 4;
                if (d < a.dh) {
                    bf16x4 w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = (bf16)(dqacc[mt][g * 4 + j] * a.scale);
                    *(bf16x4*)(row + d) = w;
                }
            }
    }
}

// dK, dV: lane owns a key, loop over query tiles; Q^T and dO^T come from LDS, the per-query log-sum-exp and
// D = rowsum(dO * O) from two LDS vectors.
template <int DHP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KC = DHP / 16, MT = (DHP + 31) / 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, n = blockIdx.z, hd = head * a.dh;
    const int SP = (a.S + 31) & ~31, stride = SP + 4;
    bf16* qtr = (bf16*)smem;
    bf16* dotr = qtr + DHP * stride;
    float* s_lse = (float*)(dotr + DHP * stride);
    float* s_d = s_lse + SP;
    const bf16* qbase = a.q + (long long)n * a.S * a.ldq + hd;
    const bf16* dobase = a.dout + (long long)n * a.S * a.ldd + hd;
    const bf16* obase = a.o + (long long)n * a.S * a.ldo + hd;
    fill_transposed<DHP>(qtr, stride, qbase, a.ldq, a.S, SP, a.dh, tid);
    fill_transposed<DHP>(dotr, stride, dobase, a.ldd, a.S, SP, a.dh, tid);
    for (int qq = tid; qq < SP; qq += 256) {
        float d = 0.f, l = INFINITY;                  // rows past the end get probability exp2(-inf) = 0
        if (qq < a.S) {
            for (int d0 = 0; d0 < a.dh; d0 += 8) {
                const bf16x8 x = load_frag(dobase + (long long)qq * a.ldd, d0, a.dh, true);
                const bf16x8 y = load_frag(obase + (long long)qq * a.ldo, d0, a.dh, true);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += (float)x[j] * (float)y[j];
            }
            l = a.lse[((long long)n * a.heads + head) * a.S + qq];
        }
        s_d[qq] = d;
        s_lse[qq] = l;
    }
    __syncthreads();
    const int key = blockIdx.x * 128 + wave * 32 + col;
    const bool kv = key < a.Sk;
    const long long ki = (long long)n * a.Sk + (kv ? key : 0);
    bf16x8 bk[KC], bv[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        bk[kc] = load_frag(a.k + ki * a.ldk + hd, kc * 16 + h * 8, a.dh, kv);
        bv[kc] = load_frag(a.v + ki * a.ldv + hd, kc * 16 + h * 8, a.dh, kv);
    }
    f32x16 dkacc[MT], dvacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dkacc[mt][i] = dvacc[mt][i] = 0.f;
    const bool drop = a.drop_thresh < 256;
    const uint64_t seed = drop ? *a.seed : 0;
    for (int qt = 0; qt < SP; qt += 32) {
        const int qr = qt + col;
        const bool qv = qr < a.S;
        const bf16* qrow = qbase + (long long)(qv ? qr : 0) * a.ldq;
        const bf16* dorow = dobase + (long long)(qv ? qr : 0) * a.ldd;
        f32x16 acc, adp;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = adp[i] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            acc = MFMA(load_frag(qrow, kc * 16 + h * 8, a.dh, qv), bk[kc], acc);
            adp = MFMA(load_frag(dorow, kc * 16 + h * 8, a.dh, qv), bv[kc], adp);
        }
        float p[16], ds[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q0 = qt + g * 8 + h * 4;
            const f32x4 l4 = *(const f32x4*)(s_lse + q0), d4 = *(const f32x4*)(s_d + q0);
            uint32_t keepb = 0xFu;
            if (drop) {
                const Philox4 r = attn_mask_block(a, seed, n * a.heads + head, q0 >> 2, key >> 2);
                keepb = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) keepb |= ((int)((word_of(r, j) >> (8 * (key & 3))) & 0xFFu) < a.drop_thresh ? 1u : 0u) << j;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = g * 4 + j;
                const float pr = __builtin_amdgcn_exp2f(acc[i] * a.scale2 - l4[j]);
                const bool keep = (keepb >> j) & 1u;
                const float sc = drop ? a.drop_scale : 1.f;
                p[i] = keep ? pr * sc : 0.f;
                ds[i] = pr * ((keep ? adp[i] * sc : 0.f) - d4[j]);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 pb, db;
#pragma unroll
            for (int j = 0; j < 8; ++j) { pb[j] = (bf16)p[8 * c + j]; db[j] = (bf16)ds[8 * c + j]; }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool ok = mt * 32 + col < DHP;
                dvacc[mt] = MFMA(lds_frag(dotr, stride, mt * 32 + col, qt + 16 * c, h, ok), pb, dvacc[mt]);
                dkacc[mt] = MFMA(lds_frag(qtr, stride, mt * 32 + col, qt + 16 * c, h, ok), db, dkacc[mt]);
            }
        }
    }
    if (kv) {
        bf16* krow = a.dk + ((long long)n * a.Sk + key) * a.lddk + hd;
        bf16* vrow = a.dv + ((long long)n * a.Sk + key) * a.lddv + hd;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = mt * 32 + g * 8 + h * 4;
                if (d < a.dh) {
                    bf16x4 wk, wv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { wk[j] = (bf16)(dkacc[mt][g * 4 + j] * a.scale); wv[j] = (bf16)dvacc[mt][g * 4 + j]; }
                    *(bf16x4*)(krow + d) = wk;
                    *(bf16x4*)(vrow + d) = wv;
                }
            }
    }
}

static int attn_prepare(AttnArgs& a, const ddpm_attn_desc* d, bool bwd, const char* what) {
    DDPM_REQUIRE(d && d->q && d->k && d->v && d->o, "%s: null pointer", what);
    DDPM_REQUIRE(d->N > 0 && d->S > 0 && d->Sk > 0 && d->heads > 0, "%s: bad sizes N=%d S=%d Sk=%d heads=%d", what, d->N, d->S, d->Sk, d->heads);
    DDPM_REQUIRE(d->dh == 4 || (d->dh % 8 == 0 && d->dh >= 8 && d->dh <= 64), "%s: head dim %d not supported (4, or a multiple of 8 up to 64)", what, d->dh);
    const int lds[] = {d->ldq, d->ldk, d->ldv, d->ldo};
    for (int v : lds) DDPM_REQUIRE(v >= d->heads * d->dh && v % (d->dh == 4 ? 4 : 8) == 0, "%s: bad row stride %d", what, v);
    DDPM_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->seed), "%s: dropout %g needs a seed and must be in [0,1)", what, d->drop_p);
    a.q = (const bf16*)d->q; a.k = (const bf16*)d->k; a.v = (const bf16*)d->v; a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv;
    a.o = (bf16*)d->o; a.ldo = d->ldo; a.lse = d->lse;
    a.dout = (const bf16*)d->dout; a.ldd = d->ldd;
    a.dq = (bf16*)d->dq; a.dk = (bf16*)d->dk; a.dv = (bf16*)d->dv; a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
    a.N = d->N; a.S = d->S; a.Sk = d->Sk; a.heads = d->heads; a.dh = d->dh;
    a.scale = 1.0f / sqrtf((float)d->dh);
    a.scale2 = a.scale * 1.4426950408889634f;
    a.drop_thresh = 256; a.drop_scale = 1.f;
    if (d->drop_p > 0.f) {
        int t = (int)lrintf((1.0f - d->drop_p) * 256.0f);
        t = t < 1 ? 1 : (t > 255 ? 255 : t);
        a.drop_thresh = t; a.drop_scale = 256.0f / (float)t;
    }
    a.seed = d->seed; a.stream_id = d->stream_id;
    if (bwd) {
        DDPM_REQUIRE(d->dout && d->dq && d->dk && d->dv && d->lse, "%s: null gradient pointer", what);
        const int l2[] = {d->ldd, d->lddq, d->lddk, d->lddv};
        for (int v : l2) DDPM_REQUIRE(v >= d->heads * d->dh && v % (d->dh == 4 ? 4 : 8) == 0, "%s: bad row stride %d", what, v);
    }
    return DDPM_OK;
}

static const int ATTN_LDS_MAX = 160 * 1024;

template <typename K>
static int attn_launch(K kernel, dim3 grid, size_t lds, const AttnArgs& a, hipStream_t stream, const char* what) {
    DDPM_REQUIRE(lds <= (size_t)ATTN_LDS_MAX, "%s: %zu bytes of LDS needed (tokens x padded head dim too large; limit %d)", what, lds, ATTN_LDS_MAX);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DDPM_REQUIRE(e == hipSuccess, "%s: cannot raise the LDS limit: %s", what, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, stream, a);
    return ddpm_check_launch(what);
}

extern "C" int ddpm_attention_forward(const ddpm_attn_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a;
    if (int rc = attn_prepare(a, d, false, "attention_forward")) return rc;
    const int dhp = a.dh <= 16 ? 16 : (a.dh <= 32 ? 32 : 64);
    const size_t lds = (size_t)dhp * (((a.Sk + 31) & ~31) + 4) * 2;
    const dim3 grid((a.S + 127) / 128, a.heads, a.N);
    if (dhp == 16) return attn_launch(attn_fwd_kernel<16>, grid, lds, a, stream, "attention_forward");
    if (dhp == 32) return attn_launch(attn_fwd_kernel<32>, grid, lds, a, stream, "attention_forward");
    return attn_launch(attn_fwd_kernel<64>, grid, lds, a, stream, "attention_forward");
}

extern "C" int ddpm_attention_backward(const ddpm_attn_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AttnArgs a;
    if (int rc = attn_prepare(a, d, true, "attention_backward")) return rc;
    const int dhp = a.dh <= 16 ? 16 : (a.dh <= 32 ? 32 : 64);
    const int SP = (a.S + 31) & ~31, SKP = (a.Sk + 31) & ~31;
    const size_t lds_q = (size_t)dhp * (SKP + 4) * 2;
    const size_t lds_kv = (size_t)2 * dhp * (SP + 4) * 2 + (size_t)SP * 8;
    const dim3 gq((a.S + 127) / 128, a.heads, a.N), gkv((a.Sk + 127) / 128, a.heads, a.N);
    int rc;
    if (dhp == 16) {
        if ((rc = attn_launch(attn_bwd_dq_kernel<16>, gq, lds_q, a, stream, "attention_backward(dq)"))) return rc;
        return attn_launch(attn_bwd_dkv_kernel<16>, gkv, lds_kv, a, stream, "attention_backward(dkv)");
    }
    if (dhp == 32) {
        if ((rc = attn_launch(attn_bwd_dq_kernel<32>, gq, lds_q, a, stream, "attention_backward(dq)"))) return rc;
        return attn_launch(attn_bwd_dkv_kernel<32>, gkv, lds_kv, a, stream, "attention_backward(dkv)");
    }
    if ((rc = attn_launch(attn_bwd_dq_kernel<64>, gq, lds_q, a, stream, "attention_backward(dq)"))) return rc;
    return attn_launch(attn_bwd_dkv_kernel<64>, gkv, lds_kv, a, stream, "attention_backward(dkv)");
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-sample normalisation of the ConditionalModule: y = GroupNorm(1, C)(act(a) + b) over all S*C elements of a sample
// (constituent_blocks.py:425-428 with act = identity, a = attention output projection, b = the module input; :431-434
// with a = Linear output, b = the first norm's output).  One workgroup per sample, two passes over the sample (the
// second read is an L2 hit), fp32 statistics.

struct SnArgs {
    const bf16* a;
    const bf16* b;
    const float* gamma;
    const float* beta;
    bf16* y;
    float* stats;              // [N][2] mean, rstd
    const bf16* dy;
    bf16* du;                  // gradient w.r.t. (act(a) + b), i.e. w.r.t. b
    bf16* da;                  // gradient w.r.t. a (may be NULL when act is the identity: it equals du)
    float* part;               // [N][3][C]: d gamma, d beta, column sums of da
    int N, S, C, act;
    float eps;
    const float* act_beta;     // swish / hardswish: the activation's learnable scalar
    float* dbeta_part;         // backward, swish / hardswish: [N] per-sample sums of da-terms of the scalar's gradient
};

constexpr int SN_THREADS = 512;

__device__ __forceinline__ float block_sum512(float v, float* s_w, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) s_w[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SN_THREADS / 64; ++w) t += s_w[w];
    return t;
}

// sum v[] over all threads with the same vector column (tid % vps, vps a power of two <= 64); result in every thread.
// s_col: LDS [SN_THREADS/64][vps*8]
__device__ __forceinline__ void block_colsum512(float (&v)[8], int vps, float* s_col, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = v[j];
        for (int o = 32; o >= vps; o >>= 1) x += __shfl_xor(x, o, 64);
        v[j] = x;
    }
    __syncthreads();
    if (lane < vps) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s_col[(wave * vps + lane) * 8 + j] = v[j];
    }
    __syncthreads();
    const int cv = lane % vps;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < SN_THREADS / 64; ++w) x += s_col[(w * vps + cv) * 8 + j];
        v[j] = x;
    }
}

__device__ __forceinline__ void sn_u(const SnArgs& a, long long e, float (&u)[8], float (&av)[8], float bsc) {
    const bf16x8 x = *(const bf16x8*)(a.a + e);
#pragma unroll
    for (int j = 0; j < 8; ++j) { av[j] = (float)x[j]; u[j] = act_fwd(av[j], a.act, bsc); }
    if (a.b) {
        const bf16x8 r = *(const bf16x8*)(a.b + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += (float)r[j];
    }
}

__global__ __launch_bounds__(SN_THREADS) void sample_norm_fwd_kernel(const SnArgs a) {
    __shared__ float s_w[SN_THREADS / 64];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int vps = a.C / 8, cv = tid % vps, prow = tid / vps, pstep = SN_THREADS / vps;
    const long long base = (long long)n * a.S * a.C + cv * 8;
    const float bsc = ACT_HAS_BETA(a.act) ? *a.act_beta : 1.f;
    float s = 0.f, ss = 0.f;
    for (int p = prow; p < a.S; p += pstep) {
        float u[8], av[8];
        sn_u(a, base + (long long)p * a.C, u, av, bsc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += u[j]; ss += u[j] * u[j]; }
    }
    s = block_sum512(s, s_w, tid);
    ss = block_sum512(ss, s_w, tid);
    const float im = 1.0f / ((float)a.S * (float)a.C);
    const float mean = s * im;
    const float rstd = rsqrtf(fmaxf(ss * im - mean * mean, 0.f) + a.eps);
    if (tid == 0 && a.stats) { a.stats[n * 2] = mean; a.stats[n * 2 + 1] = rstd; }
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float g = a.gamma[cv * 8 + j] * rstd;
        sc[j] = g; sh[j] = a.beta[cv * 8 + j] - mean * g;
    }
    for (int p = prow; p < a.S; p += pstep) {
        const long long e = base + (long long)p * a.C;
        float u[8], av[8];
        sn_u(a, e, u, av, bsc);
        bf16x8 w;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (bf16)(u[j] * sc[j] + sh[j]);
        *(bf16x8*)(a.y + e) = w;
    }
}

// With g = dy * gamma, xhat = (u - mean) rstd, L = S*C:  du = rstd (g - mean_L(g) - xhat mean_L(g xhat)),
// da = du * act'(a), d gamma_c = sum_s dy xhat, d beta_c = sum_s dy.
__global__ __launch_bounds__(SN_THREADS) void sample_norm_bwd_kernel(const SnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_col = (float*)smem;                       // [8][C]
    __shared__ float s_w[SN_THREADS / 64];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int vps = a.C / 8, cv = tid % vps, prow = tid / vps, pstep = SN_THREADS / vps;
    const long long base = (long long)n * a.S * a.C + cv * 8;
    const float mean = a.stats[n * 2], rstd = a.stats[n * 2 + 1];
    const bool hb = ACT_HAS_BETA(a.act);
    const float bsc = hb ? *a.act_beta : 1.f;
    float sb = 0.f;
    float ga[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ga[j] = a.gamma[cv * 8 + j];
    float s1 = 0.f, s2 = 0.f, dg[8], db[8], dbias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dg[j] = db[j] = dbias[j] = 0.f;
    for (int p = prow; p < a.S; p += pstep) {
        const long long e = base + (long long)p * a.C;
        float u[8], av[8];
        sn_u(a, e, u, av, bsc);
        const bf16x8 dv = *(const bf16x8*)(a.dy + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (u[j] - mean) * rstd, d = (float)dv[j], g = d * ga[j];
            s1 += g; s2 += g * xh;
            dg[j] += d * xh; db[j] += d;
        }
    }
    s1 = block_sum512(s1, s_w, tid);
    s2 = block_sum512(s2, s_w, tid);
    const float im = 1.0f / ((float)a.S * (float)a.C);
    s1 *= im; s2 *= im;
    for (int p = prow; p < a.S; p += pstep) {
        const long long e = base + (long long)p * a.C;
        float u[8], av[8];
        sn_u(a, e, u, av, bsc);
        const bf16x8 dv = *(const bf16x8*)(a.dy + e);
        bf16x8 wu, wa;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (u[j] - mean) * rstd;
            const float du = rstd * ((float)dv[j] * ga[j] - s1 - xh * s2);
            const float da = du * act_bwd(av[j], a.act, bsc);
            if (hb) sb += du * act_dbeta(av[j], a.act, bsc);
            dbias[j] += da;
            wu[j] = (bf16)du; wa[j] = (bf16)da;
        }
        *(bf16x8*)(a.du + e) = wu;
        if (a.da) *(bf16x8*)(a.da + e) = wa;
    }
    block_colsum512(dg, vps, s_col, tid);
    block_colsum512(db, vps, s_col, tid);
    block_colsum512(dbias, vps, s_col, tid);
    if (tid < vps) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a.part[((long long)n * 3 + 0) * a.C + tid * 8 + j] = dg[j];
            a.part[((long long)n * 3 + 1) * a.C + tid * 8 + j] = db[j];
            a.part[((long long)n * 3 + 2) * a.C + tid * 8 + j] = dbias[j];
        }
    }
    if (hb) {
        sb = block_sum512(sb, s_w, tid);
        if (tid == 0) a.dbeta_part[n] = sb;
    }
}

static int sn_prepare(SnArgs& a, const ddpm_sample_norm_desc* d, bool bwd, const char* what) {
    DDPM_REQUIRE(d && d->a && d->gamma && d->stats, "%s: null pointer", what);
    DDPM_REQUIRE(d->N > 0 && d->S > 0, "%s: bad sizes", what);
    const int vps = d->C / 8;
    DDPM_REQUIRE(d->C >= 8 && d->C % 8 == 0 && vps <= 64 && (vps & (vps - 1)) == 0, "%s: C (%d) must be 8 x a power of two, at most 512", what, d->C);
    a.a = (const bf16*)d->a; a.b = (const bf16*)d->b; a.gamma = d->gamma; a.beta = d->beta; a.y = (bf16*)d->y; a.stats = d->stats;
    a.dy = (const bf16*)d->dy; a.du = (bf16*)d->du; a.da = (bf16*)d->da; a.part = d->part;
    a.N = d->N; a.S = d->S; a.C = d->C; a.act = d->act; a.eps = d->eps;
    a.act_beta = d->act_beta; a.dbeta_part = d->dbeta_part;
    DDPM_REQUIRE(d->act >= 0 && d->act <= ACT_LAST, "%s: unknown activation %d", what, d->act);
    DDPM_REQUIRE(!ACT_HAS_BETA(d->act) || (d->act_beta && (!bwd || d->dbeta_part)),
                 "%s: swish / hardswish need their scalar (act_beta) and, backward, the [N] workspace of its gradient", what);
    if (!bwd) DDPM_REQUIRE(d->y && d->beta, "%s: null output / beta", what);
    else DDPM_REQUIRE(d->dy && d->du && d->part, "%s: null gradient pointer", what);
    return DDPM_OK;
}

extern "C" int ddpm_sample_norm_forward(const ddpm_sample_norm_desc* d, ddpm_stream_t stream_) {
    SnArgs a;
    if (int rc = sn_prepare(a, d, false, "sample_norm_forward")) return rc;
    hipLaunchKernelGGL(sample_norm_fwd_kernel, dim3(a.N), dim3(SN_THREADS), 0, (hipStream_t)stream_, a);
    return ddpm_check_launch("sample_norm_forward");
}

extern "C" int ddpm_sample_norm_backward(const ddpm_sample_norm_desc* d, ddpm_stream_t stream_) {
    SnArgs a;
    if (int rc = sn_prepare(a, d, true, "sample_norm_backward")) return rc;
    hipLaunchKernelGGL(sample_norm_bwd_kernel, dim3(a.N), dim3(SN_THREADS), (size_t)(SN_THREADS / 64) * a.C * 4, (hipStream_t)stream_, a);
    return ddpm_check_launch("sample_norm_backward");
}
