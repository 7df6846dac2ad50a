// All time-embedding MLPs of the denoiser in one launch (forward) and one launch (backward data path).
//
// The reference builds one EmbedBlock per encoder / decoder stage (constituent_blocks.py:282-328:
// Linear(t_dim -> 2C) -> act -> Linear(2C -> C), output broadcast-added to the stage input, unets.py:128,237).
// All eight depend only on the sinusoidal embedding of t, so they can run together before the first stage instead
// of as two dozen latency-bound launches threaded through the network; likewise their backward needs only the
// per-image sums of the stage-input gradients, all of which exist at the end of the backward pass.
//
// Workgroup = (32 rows of the batch) x (one MLP); both GEMMs on v_mfma_f32_32x32x16_bf16, hidden activations in LDS.
#include "common.h"
#include "ddpm_hip.h"
#include "sir_tile.h"

#define EMB_T 32          // width of the sinusoidal embedding the kernels are built for

struct EmbedFwdArgs {
    const bf16* sin;      // [N][32]
    int N, act, nmlp;
    int C[DDPM_EMBED_MAX];
    const bf16* w0[DDPM_EMBED_MAX]; const float* b0[DDPM_EMBED_MAX];
    const bf16* w1[DDPM_EMBED_MAX]; const float* b1[DDPM_EMBED_MAX];
    float* out[DDPM_EMBED_MAX];
    bf16* raw0[DDPM_EMBED_MAX]; bf16* h[DDPM_EMBED_MAX];
};

struct EmbedBwdArgs {
    int N, act, nmlp;
    int C[DDPM_EMBED_MAX];
    const float* dtemb[DDPM_EMBED_MAX]; const bf16* raw0[DDPM_EMBED_MAX]; const bf16* w1d[DDPM_EMBED_MAX];
    bf16* dt16[DDPM_EMBED_MAX]; bf16* dz[DDPM_EMBED_MAX];
    float* db1[DDPM_EMBED_MAX]; float* db0[DDPM_EMBED_MAX];
};

// 32x32 tile of A[32][K] (LDS, row stride AS bytes) x B[n0..n0+31][K]^T (global bf16, row stride K), K at run time
__device__ __forceinline__ f32x16 tile_rt(const char* s_a, int AS, const bf16* wrow, int K, int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const char* ap = s_a + (lane & 31) * AS + (lane >> 5) * 16;
    const bf16* bp = wrow + (lane >> 5) * 8;
    for (int k0 = 0; k0 < K; k0 += 64) {               // K is a multiple of 32; chunks of 4 k-steps, tail of 2
        bf16x8 bv[4];
        const int steps = (K - k0) >= 64 ? 4 : 2;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < steps) bv[u] = *(const bf16x8*)(bp + k0 + u * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < steps) {
                const bf16x8 av = *(const bf16x8*)(ap + (k0 + u * 16) * 2);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[u], acc, 0, 0, 0);
            }
    }
    return acc;
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbedFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = blockIdx.y, C = a.C[m], H = 2 * C;
    const int n0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int SS = EMB_T * 2 + 16;
    const int HS = H * 2 + 16;
    char* s_sin = smem;                 // [32][SS]
    char* s_h = smem + 32 * SS;         // [32][HS]
    if (tid < 32 * (EMB_T / 8)) {
        const int row = tid / (EMB_T / 8), v = tid % (EMB_T / 8);
        bf16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (bf16)0.f;
        if (n0 + row < a.N) x = *(const bf16x8*)(a.sin + (long long)(n0 + row) * EMB_T + v * 8);
        *(bf16x8*)(s_sin + row * SS + v * 16) = x;
    }
    __syncthreads();
    const bf16* w0 = a.w0[m];
    const float* b0 = a.b0[m];
    bf16* raw0 = a.raw0[m];
    bf16* hs = a.h[m];
    for (int nt = wave; nt < H / 32; nt += 4) {
        const int col = nt * 32 + (lane & 31);
        const f32x16 acc = tile_rt(s_sin, SS, w0 + (long long)col * EMB_T, EMB_T, lane);
        const float b = b0[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const bf16 z = (bf16)(acc[r] + b);
            const bf16 y = (bf16)act_fwd((float)z, a.act);
            *(bf16*)(s_h + row * HS + col * 2) = y;
            if (raw0 && blockIdx.z == 0 && n0 + row < a.N) {
                raw0[(long long)(n0 + row) * H + col] = z;
                hs[(long long)(n0 + row) * H + col] = y;
            }
        }
    }
    __syncthreads();
    const bf16* w1 = a.w1[m];
    const float* b1 = a.b1[m];
    float* out = a.out[m];
    // the second GEMM (K = 2C, its weight rows streamed from L2) is the long part: its column tiles are dealt over the
    // gridDim.z workgroups of this (row block, MLP), each of which has computed the cheap first GEMM for itself
    for (int nt = wave + 4 * blockIdx.z; nt < C / 32; nt += 4 * gridDim.z) {
        const int col = nt * 32 + (lane & 31);
        const f32x16 acc = tile_rt(s_h, HS, w1 + (long long)col * H, H, lane);
        const float b = b1[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (n0 + row < a.N) out[(long long)(n0 + row) * C + col] = acc[r] + b;
        }
    }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const EmbedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = blockIdx.y, C = a.C[m], H = 2 * C;
    const int n0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CS = C * 2 + 16, HS = H * 2 + 16;
    char* s_d = smem;                    // [32][CS] gradient of the MLP output, bf16
    char* s_z = smem + 32 * CS;          // [32][HS] gradient of the hidden pre-activation, bf16
    // the hidden layer's column tiles (the long part: K = C, weight rows streamed from L2) are dealt over the gridDim.z
    // workgroups of this (row block, MLP); each of them stages the output gradient for itself
    const int gz = gridDim.z, bz = blockIdx.z;
    if (4 * bz >= H / 32) return;               // a narrow MLP in a launch sized for a wide one: nothing left for this workgroup
    const float* dt = a.dtemb[m];
    bf16* dt16 = a.dt16[m];
    for (int i = tid; i < 32 * C; i += 256) {
        const int row = i / C, c = i % C;
        const float v = (n0 + row < a.N) ? dt[(long long)(n0 + row) * C + c] : 0.f;
        const bf16 b = (bf16)v;
        *(bf16*)(s_d + row * CS + c * 2) = b;
        if (bz == 0 && n0 + row < a.N) dt16[(long long)(n0 + row) * C + c] = b;
    }
    __syncthreads();
    if (bz == 0)
        for (int c = tid; c < C; c += 256) {    // bias gradient of the second linear layer: column sums of this row tile
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += (float)*(const bf16*)(s_d + r * CS + c * 2);
            a.db1[m][(long long)blockIdx.x * C + c] = s;
        }
    const bf16* w1d = a.w1d[m];
    const bf16* raw0 = a.raw0[m];
    bf16* dz = a.dz[m];
    for (int nt = wave + 4 * bz; nt < H / 32; nt += 4 * gz) {
        const int col = nt * 32 + (lane & 31);
        const f32x16 acc = tile_rt(s_d, CS, w1d + (long long)col * C, C, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            bf16 g = (bf16)0.f;
            if (n0 + row < a.N) {
                const float z = (float)raw0[(long long)(n0 + row) * H + col];
                g = (bf16)((float)(bf16)acc[r] * act_bwd(z, a.act));
                dz[(long long)(n0 + row) * H + col] = g;
            }
            *(bf16*)(s_z + row * HS + col * 2) = g;
        }
    }
    __syncthreads();
    for (int c = tid; c < H; c += 256) {        // bias gradient of the first linear layer (the columns this workgroup computed)
        if (((c >> 7) % gz) != bz) continue;
        float s = 0.f;
        for (int r = 0; r < 32; ++r) s += (float)*(const bf16*)(s_z + r * HS + c * 2);
        a.db0[m][(long long)blockIdx.x * H + c] = s;
    }
}

static int embed_check(int N, int t_dim, int act, int nmlp, const int* C) {
    DDPM_REQUIRE(N > 0 && nmlp > 0 && nmlp <= DDPM_EMBED_MAX, "embed: need 1..%d MLPs and a positive batch", DDPM_EMBED_MAX);
    DDPM_REQUIRE(t_dim == EMB_T, "embed: the fused kernels are built for t_embed_dim %d, got %d", EMB_T, t_dim);
    DDPM_REQUIRE(act >= ACT_NONE && act <= ACT_LAST && !ACT_HAS_BETA(act), "embed: bad activation %d (swish / hardswish are not fused here)", act);
    for (int i = 0; i < nmlp; ++i) DDPM_REQUIRE(C[i] >= 32 && C[i] <= 512 && C[i] % 32 == 0, "embed: width %d must be a multiple of 32 in 32..512", C[i]);
    return DDPM_OK;
}

extern "C" int ddpm_embed_forward(const ddpm_embed_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(d && d->sin, "embed_forward: null pointer");
    if (int rc = embed_check(d->N, d->t_dim, d->act, d->nmlp, d->C)) return rc;
    EmbedFwdArgs a;
    a.sin = (const bf16*)d->sin; a.N = d->N; a.act = d->act; a.nmlp = d->nmlp;
    int cmax = 0;
    for (int i = 0; i < d->nmlp; ++i) {
        DDPM_REQUIRE(d->w0[i] && d->b0[i] && d->w1[i] && d->b1[i] && d->out[i], "embed_forward: null pointer in MLP %d", i);
        DDPM_REQUIRE((d->raw0[i] != nullptr) == (d->h[i] != nullptr), "embed_forward: give both saved tensors of MLP %d or neither", i);
        a.C[i] = d->C[i]; a.w0[i] = (const bf16*)d->w0[i]; a.b0[i] = d->b0[i]; a.w1[i] = (const bf16*)d->w1[i]; a.b1[i] = d->b1[i];
        a.out[i] = d->out[i]; a.raw0[i] = (bf16*)d->raw0[i]; a.h[i] = (bf16*)d->h[i];
        if (d->C[i] > cmax) cmax = d->C[i];
    }
    const size_t lds = 32 * (EMB_T * 2 + 16) + 32 * (size_t)(4 * cmax + 16);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)embed_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ddpm_check_launch("embed_forward: raise dynamic LDS limit");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((d->N + 31) / 32, d->nmlp, cmax >= 256 ? 8 : cmax >= 128 ? 4 : 1), dim3(256), lds, stream, a);
    return ddpm_check_launch("embed_forward");
}

extern "C" int ddpm_embed_backward(const ddpm_embed_bwd_desc* d, ddpm_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DDPM_REQUIRE(d, "embed_backward: null pointer");
    if (int rc = embed_check(d->N, EMB_T, d->act, d->nmlp, d->C)) return rc;
    EmbedBwdArgs a;
    a.N = d->N; a.act = d->act; a.nmlp = d->nmlp;
    int cmax = 0;
    for (int i = 0; i < d->nmlp; ++i) {
        DDPM_REQUIRE(d->dtemb[i] && d->raw0[i] && d->w1d[i] && d->dt16[i] && d->dz[i] && d->db1[i] && d->db0[i],
                     "embed_backward: null pointer in MLP %d", i);
        a.C[i] = d->C[i]; a.dtemb[i] = d->dtemb[i]; a.raw0[i] = (const bf16*)d->raw0[i]; a.w1d[i] = (const bf16*)d->w1d[i];
        a.dt16[i] = (bf16*)d->dt16[i]; a.dz[i] = (bf16*)d->dz[i]; a.db1[i] = d->db1[i]; a.db0[i] = d->db0[i];
        if (d->C[i] > cmax) cmax = d->C[i];
    }
    const size_t lds = 32 * (size_t)(2 * cmax + 16) + 32 * (size_t)(4 * cmax + 16);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)embed_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ddpm_check_launch("embed_backward: raise dynamic LDS limit");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((d->N + 31) / 32, d->nmlp, cmax >= 256 ? 4 : cmax >= 128 ? 2 : 1), dim3(256), lds, stream, a);
    return ddpm_check_launch("embed_backward");
}
