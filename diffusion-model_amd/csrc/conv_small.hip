// 3x3 stride-1 convolution (forward or data-gradient form) on images of 2x2 or 4x4 pixels.
//
// The deepest UNet stages run 3x3 convolutions with 128-512 channels on a few pixels per image: a GEMM with only
// N*H*W = 1-4 K rows but K = 9*Cin up to 4608 deep.  The general tap-list kernel walks K in 32-channel chunks with a
// stage/barrier/MFMA round per chunk and has too few workgroups to hide that; here a workgroup owns a 32-row x 32-column
// output tile, keeps the input rows of its images in LDS once (plus one zero row that all out-of-image taps read), and
// its four waves split the input channels four ways, each streaming its weight fragments straight from L2 with a tap
// of look-ahead; the four partial tiles are added through LDS.  Same operands, fused input transforms (channel concat,
// time-embedding add) and epilogue (bias, activation, residual, per-image sums) as the general kernel.
#include "conv_geom.h"
#include "ddpm_hip.h"

struct SmallConvArgs {
    const bf16* in0; const bf16* in1; int C0, C1;
    const float* in_add;
    const bf16* w; const float* bias;
    bf16* out; const bf16* res;
    float* sum_out;
    int N, Cout, out_cs, w_rows, act;
};

template <int W>
__global__ __launch_bounds__(256) void conv_small_kernel(const SmallConvArgs a) {
    constexpr int HW = W * W;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cin = a.C0 + a.C1;
    const int AS = Cin * 2 + 16;
    char* s_a = smem;                                  // [33][AS]; row 32 stays zero
    float* s_red = (float*)(smem + 33 * AS);           // [4][32][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row0 = (long long)blockIdx.x * 32;
    const long long rows_total = (long long)a.N * HW;
    const int co0 = blockIdx.y * 32;

    // ---- input rows of this tile's images -> LDS (concat of two sources; broadcast add on source 0)
    const int vpr = Cin / 8;
    for (int idx = tid; idx < 33 * vpr; idx += 256) {
        const int row = idx / vpr, c = (idx - row * vpr) * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)0.f;
        if (row < 32 && row0 + row < rows_total) {
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
    const int kq = Cin / 4;                            // channels per wave (multiple of 16)
    const int ksteps = kq / 16;                        // <= 8
    const int r = lane & 31;
    const int p = r % HW, py = p / W, px = p % W;
    const bf16* wbase = a.w + ((long long)(co0 + r) * Cin + wave * kq + (lane >> 5) * 8);
    const long long slice = (long long)a.w_rows * Cin;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 bcur[8], bnext[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        if (u < ksteps) bcur[u] = *(const bf16x8*)(wbase + u * 16);
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < ksteps) bnext[u] = *(const bf16x8*)(wbase + (tap + 1) * slice + u * 16);
        }
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int ny = py + dy, nx = px + dx;
        const int nrow = (ny >= 0 && ny < W && nx >= 0 && nx < W) ? r + dy * W + dx : 32;
        const char* ap = s_a + nrow * AS + (wave * kq + (lane >> 5) * 8) * 2;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < ksteps) {
                const bf16x8 av = *(const bf16x8*)(ap + u * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bcur[u], acc, 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u) bcur[u] = bnext[u];
    }
    // ---- add the four partial tiles
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        s_red[(wave * 32 + row) * 33 + (lane & 31)] = acc[i];
    }
    __syncthreads();
    {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const long long grow = row0 + row;
        if (grow < rows_total) {
            bf16x4 o;
            float sums[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + c4 + j;
                float v = s_red[(0 * 32 + row) * 33 + c4 + j] + s_red[(1 * 32 + row) * 33 + c4 + j] +
                          s_red[(2 * 32 + row) * 33 + c4 + j] + s_red[(3 * 32 + row) * 33 + c4 + j];
                v = act_fwd(v + ((a.bias && co < a.Cout) ? a.bias[co] : 0.f), a.act);
                sums[j] = v;
                o[j] = (bf16)v;
            }
            if (a.sum_out) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co0 + c4 + j < a.Cout) atomicAdd(a.sum_out + (grow / HW) * a.Cout + co0 + c4 + j, sums[j]);
            }
            const long long e = grow * a.out_cs + co0 + c4;
            if (a.res) {
                const bf16x4 rv = *(const bf16x4*)(a.res + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)((float)o[j] + (float)rv[j]);
            }
            if (co0 + c4 + 4 <= a.Cout) {
                *(bf16x4*)(a.out + e) = o;
            } else {
                for (int j = 0; j < a.Cout - co0 - c4; ++j) a.out[e + j] = o[j];
            }
        }
    }
}

// Returns 1 when the problem was launched here, 0 when the caller should use the general kernel, < 0 on error.
int ddpm_conv_small_try(const ddpm_conv_desc* d, hipStream_t stream) {
    if (d->kind != CONV_3X3_S1 || d->IH != d->IW || (d->IW != 2 && d->IW != 4)) return 0;
    if (d->out_mode != OUT_BF16_NHWC || d->in_mask || d->out_mask) return 0;
    const int Cin = d->C0 + d->C1;
    if (Cin % 64 || Cin > 512 || d->C0 % 8 || d->C1 % 8 || d->out_cs % 4) return 0;
    SmallConvArgs a;
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.C0 = d->C0; a.C1 = d->C1; a.in_add = d->in_add;
    a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (bf16*)d->out; a.res = (const bf16*)d->res; a.sum_out = d->sum_out;
    a.N = d->N; a.Cout = d->Cout; a.out_cs = d->out_cs; a.w_rows = d->w_rows; a.act = d->act;
    const size_t lds = (size_t)33 * (Cin * 2 + 16) + 4 * 32 * 33 * 4;
    const long long rows = (long long)d->N * d->IH * d->IW;
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)(d->CoutW / 32));
    if (d->IW == 2) {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_small_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ddpm_check_launch("conv_small: raise dynamic LDS limit") ? -1 : -1;
        hipLaunchKernelGGL(conv_small_kernel<2>, grid, dim3(256), lds, stream, a);
    } else {
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)conv_small_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ddpm_check_launch("conv_small: raise dynamic LDS limit") ? -1 : -1;
        hipLaunchKernelGGL(conv_small_kernel<4>, grid, dim3(256), lds, stream, a);
    }
    return ddpm_check_launch("conv_small") == DDPM_OK ? 1 : -1;
}
