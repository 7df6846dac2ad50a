// Matrix-core-bound convolutions: 3x3 stride 1 on whole images of 4x4 / 8x8 / 16x16 pixels, and 1x1, with >= 64 input and
// >= 128 output channels, at batch sizes where M = N*H*W fills the machine (the UNet's 128->128 at 16x16, 256->256 at 8x8,
// 256->512 / 512->256 at 4x4 and the 1x1 expansions / projections of those stages: unets.py:78,160,
// constituent_blocks.py:81,167, conv_block.py:59-65,102-108 of the reference; their data-gradients run here too, as the
// same convolution on the flipped / transposed weight copy).
//
// The general tap-list kernel (conv_mfma.hip) stages 32 input channels at a time behind two barriers and exposes the global
// latency once per chunk; it is sized for the latency-bound launches of small batches.  This kernel is the dense contraction
// done for the matrix pipes:
//   * workgroup = 512 threads = 8 waves as 4 (M) x 2 (N), tile = 256 positions x 128 output channels, every wave owns
//     64 x 64 = 2 x 2 accumulator tiles of v_mfma_f32_32x32x16_bf16: each LDS fragment read feeds two MFMAs;
//   * the 256 positions are whole images (1, 4 or 16 of them), so the input patch of a 64-channel chunk is staged in LDS
//     ONCE for all nine taps: a tap reads the neighbour pixel's row, or a row of zeros outside the image;
//   * the loader fuses the channel concat of two sources and the broadcast add of the time embedding;
//   * K runs as a flat sequence of steps (64-channel chunk, tap); the weight tile of a step ([128][64], 16 KB) sits in a ring
//     of three LDS buffers, fetched from L2 into registers three steps ahead and written to LDS two steps ahead; the next
//     chunk's patch is fetched and written the same way into the second patch buffer.  One barrier per step, 16 MFMAs per
//     wave between barriers, and no wave ever waits on a load it has just issued.
#include "conv_geom.h"
#include "ddpm_hip.h"

namespace {

constexpr int PT = 512;                         // threads
constexpr int BM = 256, BN = 128, CKP = 64;     // tile: positions x output channels; channels per chunk
constexpr int RS = CKP * 2 + 16;                // LDS row stride of patch and weight rows (144 B: conflict-free fragment reads)
constexpr int P_BYTES = BM * RS;                // one patch buffer
constexpr int W_BYTES = BN * RS;                // one weight buffer
constexpr int ZOFF = 2 * P_BYTES + 3 * W_BYTES; // the zero row
constexpr int LDS_BYTES = ZOFF + 256;
constexpr int OS = BN * 2 + 16;                 // epilogue row stride
static_assert(BM * OS <= 2 * P_BYTES + 3 * W_BYTES, "the output tile reuses the operand buffers");

struct PipeArgs {
    const bf16* in0; const bf16* in1; int C0, C1;
    const float* in_add;                        // optional fp32 [N][C0]: broadcast add on source 0 (the stage-input time embedding)
    const bf16* w; const float* bias;
    bf16* out; const bf16* res;
    long long rows_total;                       // N * H * W
    int W, lw, hw, lhw;                         // image width (0 for 1x1: no neighbours), log2, pixels per image, log2
    int ntaps, w_rows, Cout, out_cs, act;
};

// NTAPS = 9 (3x3) or 1 (1x1).  The K loop is unrolled over groups of U steps so that every index that selects a register set
// or a tap is a compile-time constant: U is a multiple of 3 (the weight ring and its three register sets) and, for 1x1, of 2
// (the two patch buffers).  Loads past the last step are clamped to it instead of branched around (a redundant tile that
// nobody reads): with loads inside branches the compiler can no longer count which of them a wait is for, and waits for all.
template <int NTAPS, bool HAS_ADD>
__global__ __launch_bounds__(PT) void conv_pipe_kernel(const PipeArgs a) {
    constexpr int U = NTAPS == 9 ? 9 : 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wm = wave & 3, wn = wave >> 2;
    const long long row0 = (long long)blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;
    const int Cin = a.C0 + a.C1, NC = Cin / CKP, S = NC * NTAPS;
    char* s_w = smem + 2 * P_BYTES;

    // ---- staging roles: a thread moves 4 16-byte pieces of a patch chunk and 2 of a weight tile
    const int prow = tid >> 3, part = tid & 7;                                // pieces: row = prow + 64 u, 8 pieces per row
    bool rok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rok[u] = row0 + prow + 64 * u < a.rows_total;
    // image of each of this thread's four patch rows (for the broadcast add); rows of one thread are 64 apart
    long long pimg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) pimg[u] = (row0 + prow + 64 * u) >> a.lhw;
    auto load_patch = [&](bf16x8 (&v)[4], int c) {
        const int ch = c * CKP;
        const bool s1 = ch >= a.C0;
        const bf16* src = s1 ? a.in1 : a.in0;
        const int cs = s1 ? a.C1 : a.C0, cc = (s1 ? ch - a.C0 : ch) + part * 8;
        const bf16* p = src + (row0 + prow) * cs + cc;
        const bool add = HAS_ADD && !s1;
        f32x4 ad[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // rows past the end of the tensor read the tile's first row (always valid) and are zeroed: no branch around a load
            v[u] = *(const bf16x8*)(rok[u] ? p + (long long)64 * u * cs : src + row0 * cs + cc);
            if constexpr (HAS_ADD) {
                const float* ap = a.in_add + (rok[u] ? pimg[u] : (row0 >> a.lhw)) * a.C0 + (s1 ? 0 : cc);
                ad[u][0] = *(const f32x4*)ap; ad[u][1] = *(const f32x4*)(ap + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (HAS_ADD) {
                if (add) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[u][j] = (bf16)((float)v[u][j] + ad[u][0][j]); v[u][4 + j] = (bf16)((float)v[u][4 + j] + ad[u][1][j]); }
                }
            }
            if (!rok[u]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][j] = (bf16)0.f;
            }
        }
    };
    auto store_patch = [&](const bf16x8 (&v)[4], int buf) {
        char* p = smem + buf * P_BYTES + part * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) *(bf16x8*)(p + (prow + 64 * u) * RS) = v[u];
    };
    const bf16* wbase = a.w + ((long long)co0 + prow) * Cin + part * 8;
    const long long wslice = (long long)a.w_rows * Cin;
    auto load_w = [&](bf16x8 (&v)[2], int c, int tap) {
        const bf16* wp = wbase + tap * wslice + c * CKP;
#pragma unroll
        for (int u = 0; u < 2; ++u) v[u] = *(const bf16x8*)(wp + (long long)64 * u * Cin);
    };
    auto store_w = [&](const bf16x8 (&v)[2], int slot) {
        char* p = s_w + slot * W_BYTES + part * 16;
#pragma unroll
        for (int u = 0; u < 2; ++u) *(bf16x8*)(p + (prow + 64 * u) * RS) = v[u];
    };
    // weight tile of flat step s (clamped to the last one)
    auto load_w_step = [&](bf16x8 (&v)[2], int s) {
        s = s < S ? s : S - 1;
        if constexpr (NTAPS == 9) { const int c = s / 9; load_w(v, c, s - c * 9); }
        else load_w(v, s, 0);
    };

    // ---- this lane's two A rows (positions) and their pixel coordinates
    int arow[2], apy[2], apx[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        arow[mt] = wm * 64 + mt * 32 + r;
        const int p = arow[mt] & (a.hw - 1);
        apy[mt] = NTAPS == 9 ? p >> a.lw : 0;
        apx[mt] = NTAPS == 9 ? p & (a.W - 1) : 0;
    }
    const int boff = (wn * 64 + r) * RS + h * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // Fragment addresses of a step: the two A rows of this lane under the step's tap, and its first B row.
    auto step_ptrs = [&](int pbuf, int slot, int tap, const char* (&ap)[2], const char*& wb) {
        const char* pb = smem + pbuf * P_BYTES;
        wb = s_w + slot * W_BYTES + boff;
        const int dy = NTAPS == 9 ? tap / 3 - 1 : 0, dx = NTAPS == 9 ? tap % 3 - 1 : 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int ny = apy[mt] + dy, nx = apx[mt] + dx;
            const bool ok = NTAPS == 1 || (ny >= 0 && ny < a.W && nx >= 0 && nx < a.W);
            ap[mt] = ok ? pb + (arow[mt] + dy * a.W + dx) * RS + h * 16 : smem + ZOFF + h * 16;
        }
    };
    // The fragments of a step's first k-step can be read BEFORE the barrier that ends the previous step (what they read was
    // published one barrier earlier): the matrix pipe then has work the moment the barrier opens instead of waiting for eight
    // waves' LDS reads to drain.  f = {a0, a1, b0, b1} of k-step 0, or fetched here when PRE is false.
    auto compute = [&](int pbuf, int slot, int tap, bf16x8 (&f)[4], bool pre) {
        const char* ap[2];
        const char* wb;
        step_ptrs(pbuf, slot, tap, ap, wb);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x8 a0, a1, b0, b1;
            if (k == 0 && pre) { a0 = f[0]; a1 = f[1]; b0 = f[2]; b1 = f[3]; }
            else {
                a0 = *(const bf16x8*)(ap[0] + k * 32);               // (the zero row is as long as a patch row)
                a1 = *(const bf16x8*)(ap[1] + k * 32);
                b0 = *(const bf16x8*)(wb + k * 32);
                b1 = *(const bf16x8*)(wb + 32 * RS + k * 32);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    auto prefetch = [&](int pbuf, int slot, int tap, bf16x8 (&f)[4]) {
        const char* ap[2];
        const char* wb;
        step_ptrs(pbuf, slot, tap, ap, wb);
        f[0] = *(const bf16x8*)ap[0]; f[1] = *(const bf16x8*)ap[1];
        f[2] = *(const bf16x8*)wb; f[3] = *(const bf16x8*)(wb + 32 * RS);
    };

    // ---- prologue: zero row, patch of chunk 0, weight tiles of steps 0 and 1 in LDS; step 2's in registers (set 2)
    if (tid < 64) ((float*)(smem + ZOFF))[tid] = 0.f;
    bf16x8 pr[4], wr[3][2];
    load_patch(pr, 0);
    load_w_step(wr[0], 0);
    load_w_step(wr[1], 1);
    store_patch(pr, 0);
    store_w(wr[0], 0);
    store_w(wr[1], 1);
    load_w_step(wr[2], 2);
    __syncthreads();
    bf16x8 fr[4];
    if constexpr (NTAPS == 9) prefetch(0, 0, 0, fr);

    // Step s = g U + t: the weight tile of step s + 3 is fetched into register set t % 3, the tile of step s + 2 (fetched one
    // step ago, set (t + 2) % 3) is written to ring slot (t + 2) % 3 after the multiplications, one barrier ends the step.
    // 3x3: a group is one 64-channel chunk, t its tap; the next chunk's patch is fetched at tap 0 and written at tap 2.
    // 1x1: every step is a chunk of its own; the next one's patch is fetched and written within the step.
    const int G = (S + U - 1) / U;
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const int s = g * U + t;
            load_w_step(wr[t % 3], s + 3);
            if constexpr (NTAPS == 9) {
                if (t == 0) load_patch(pr, g + 1 < NC ? g + 1 : NC - 1);
                compute(g & 1, t % 3, t, fr, true);
            } else {
                load_patch(pr, s + 1 < NC ? s + 1 : NC - 1);
                if (s < S) compute(t & 1, t % 3, 0, fr, false);
            }
            store_w(wr[(t + 2) % 3], (t + 2) % 3);
            if constexpr (NTAPS == 9) {
                if (t == 2) store_patch(pr, (g + 1) & 1);
                // first fragments of the next step (its weight slot was written two steps ago, its patch at tap 2 of the
                // previous chunk at the latest: both published by barriers already passed)
                if (t + 1 < U) prefetch(g & 1, (t + 1) % 3, t + 1, fr);
                else prefetch((g + 1) & 1, 0, 0, fr);
            } else {
                store_patch(pr, (t + 1) & 1);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: bias + activation -> bf16 tile in LDS -> 16-byte NHWC stores (+ residual)
    char* s_out = smem;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = wn * 64 + nt * 32 + r;
        const float b = a.bias ? a.bias[co0 + col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (a.act == ACT_NONE) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    *(bf16*)(s_out + (wm * 64 + mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * OS + col * 2) = (bf16)(acc[mt][nt][q] + b);
            } else if (a.act == ACT_GELU) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    *(bf16*)(s_out + (wm * 64 + mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * OS + col * 2) = (bf16)act_fwd(acc[mt][nt][q] + b, ACT_GELU);
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    *(bf16*)(s_out + (wm * 64 + mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * OS + col * 2) = (bf16)act_fwd(acc[mt][nt][q] + b, a.act);
            }
        }
    }
    __syncthreads();
    const int opart = tid & 15;
    for (int row = tid >> 4; row < BM; row += PT / 16) {
        const long long grow = row0 + row;
        if (grow >= a.rows_total) break;
        bf16x8 v = *(const bf16x8*)(s_out + row * OS + opart * 16);
        const long long e = grow * a.out_cs + co0 + opart * 8;
        if (a.res) {
            const bf16x8 rv = *(const bf16x8*)(a.res + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)rv[j]);
        }
        *(bf16x8*)(a.out + e) = v;
    }
}

thread_local int g_pipe_mode = 0;           // 0: by problem size; 1: always where the shape allows; -1: never

}  // namespace

// Test / measurement hook (no counterpart in the reference): 1 = take this kernel wherever the shape allows (small test
// problems would otherwise never reach it), -1 = never, 0 = by problem size.  Returns the previous setting.
extern "C" int ddpm_conv_pipe_mode(int mode) {
    const int prev = g_pipe_mode;
    if (mode >= -1 && mode <= 1) g_pipe_mode = mode;
    return prev;
}

int ddpm_conv_pipe_mode_now() { return g_pipe_mode; }

// Returns 1 when the problem was launched here, 0 when the caller should use another kernel, < 0 on error.
int ddpm_conv_pipe_try(const ddpm_conv_desc* d, hipStream_t stream) {
    if (g_pipe_mode < 0) return 0;
    if (d->kind != CONV_3X3_S1 && d->kind != CONV_1X1) return 0;
    if (d->out_mode != OUT_BF16_NHWC || d->in_mask || d->out_mask || d->sum_out || d->out2 || d->gn_gamma || d->gnb_raw || d->gn_part ||
        d->in_affine || d->C0_used || !d->out)
        return 0;
    if (d->C0 % CKP || d->C1 % CKP || d->CoutW % BN || d->Cout != d->CoutW || d->out_cs % 8) return 0;
    if (d->act != ACT_NONE && d->act != ACT_GELU && d->act != ACT_SILU && d->act != ACT_RELU && d->act != ACT_TANH) return 0;
    const long long rows = (long long)d->N * d->IH * d->IW;
    int W = 0;
    if (d->kind == CONV_3X3_S1) {
        if (d->IH != d->IW || (d->IW != 4 && d->IW != 8 && d->IW != 16)) return 0;
        W = d->IW;
    }
    // by size: worth it once the launch has a workgroup per CU and K is deep enough to amortise the pipeline's fill
    if (g_pipe_mode == 0 && (rows * (d->CoutW / BN) < 256ll * BM || (d->C0 + d->C1) * (W ? 9 : 1) < 512)) return 0;
    PipeArgs a;
    a.in0 = (const bf16*)d->in0; a.in1 = (const bf16*)d->in1; a.C0 = d->C0; a.C1 = d->C1;
    a.in_add = d->in_add;
    a.w = (const bf16*)d->w; a.bias = d->bias; a.out = (bf16*)d->out; a.res = (const bf16*)d->res;
    a.rows_total = rows;
    a.W = W; a.hw = W ? W * W : 1; a.lw = 0; a.lhw = 0;
    while ((1 << a.lw) < W) ++a.lw;
    a.lhw = 2 * a.lw;
    if (!W) {                                   // 1x1: the image index of a row (only the broadcast add needs it)
        const int hw = d->IH * d->IW;
        if (d->in_add && (hw & (hw - 1))) return 0;
        while ((1 << a.lhw) < hw) ++a.lhw;
    }
    a.ntaps = W ? 9 : 1; a.w_rows = d->w_rows; a.Cout = d->Cout; a.out_cs = d->out_cs; a.act = d->act;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_pipe_kernel<9, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv_pipe_kernel<9, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv_pipe_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv_pipe_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            ddpm_set_error("conv_pipe: cannot reserve %d bytes of LDS", LDS_BYTES);
            return -1;
        }
        attr = true;
    }
    dim3 grid((unsigned)((rows + BM - 1) / BM), (unsigned)(d->CoutW / BN));
    if (W && a.in_add) hipLaunchKernelGGL((conv_pipe_kernel<9, true>), grid, dim3(PT), LDS_BYTES, stream, a);
    else if (W) hipLaunchKernelGGL((conv_pipe_kernel<9, false>), grid, dim3(PT), LDS_BYTES, stream, a);
    else if (a.in_add) hipLaunchKernelGGL((conv_pipe_kernel<1, true>), grid, dim3(PT), LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((conv_pipe_kernel<1, false>), grid, dim3(PT), LDS_BYTES, stream, a);
    return ddpm_check_launch("conv_pipe") == DDPM_OK ? 1 : -1;
}
