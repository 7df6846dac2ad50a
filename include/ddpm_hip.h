/* libddpm_hip.so - C ABI of the MI355X (gfx950) DDPM hot path.
 *
 * The reference project is pure Python on PyTorch and has no FFI of its own; the seam this library
 * plugs into is the torch-op level underneath its nn.Modules.  Each entry point below names the
 * reference code whose arithmetic it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless a comment says "host";
 *   - activations are NHWC bf16 (channel count a multiple of 8, 16 for convolution inputs);
 *     the module boundary tensors (images, noise, predictions) are NCHW fp32 as in the reference;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and nothing synchronises;
 *   - no entry point allocates device memory;
 *   - return value 0 = success, <0 = error (ddpm_last_error() gives the message; thread-local).
 */
#ifndef DDPM_HIP_H
#define DDPM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ddpm_stream_t;

/* activation codes (reference neural_nets/activations.py:102-114; only parameter-free ones are fused) */
enum { DDPM_ACT_NONE = 0, DDPM_ACT_GELU = 1, DDPM_ACT_SILU = 2, DDPM_ACT_RELU = 3, DDPM_ACT_TANH = 4, DDPM_ACT_SIGMOID = 5 };

/* convolution kinds handled by ddpm_conv_forward / ddpm_conv_wgrad */
enum {
    DDPM_CONV_3X3_S1 = 0,     /* nn.Conv2d(k=3, p=1)            unets.py:50,78,160,179,182; constituent_blocks.py:81,167 */
    DDPM_CONV_3X3_S2 = 1,     /* nn.Conv2d(k=3, s=2, p=1)       constituent_blocks.py:66 */
    DDPM_CONV_1X1 = 2,        /* nn.Conv2d(k=1) and nn.Linear   conv_block.py:59-65,102-108; constituent_blocks.py:309,313 */
    DDPM_CONVT_2X2_S2 = 3,    /* nn.ConvTranspose2d(k=2, s=2)   constituent_blocks.py:180 */
    DDPM_DGRAD_3X3_S2 = 4,    /* data-gradient of DDPM_CONV_3X3_S2 */
    DDPM_DGRAD_CONVT_2X2 = 5  /* data-gradient of DDPM_CONVT_2X2_S2 */
};
enum { DDPM_OUT_BF16_NHWC = 0, DDPM_OUT_F32_NCHW = 1 };

const char* ddpm_last_error(void);
int ddpm_abi_version(void);

/* ---- diffusion process (elementwise.hip) ------------------------------------------------------ */

/* Per-timestep table `sched`: fp32 [T+1][8], row t = { sqrt(abar_t), sqrt(1-abar_t), A_t, B_t, C_t, w_t, 0, 0 }
 * built by the host from the reference's schedule tensors (ddpm.py:44-62, ELBO.py:66-67). */

/* DDPM.q (ddpm.py:81-106): x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps.  x0: NCHW fp32.  noise: NCHW fp32, or
 * NULL to draw eps in-kernel from Philox keyed by *seed and stream_id.  Writes xt (NCHW fp32) and/or xt_nhwc
 * (NHWC bf16 with cpad channels, padding zeroed); either may be NULL.  H*W must be a multiple of 4. */
int ddpm_q_sample(const float* x0, const int64_t* t, const float* noise, const float* sched, float* xt, void* xt_nhwc,
                  int cpad, int N, int C, int HW, const uint64_t* seed, uint32_t stream_id, ddpm_stream_t stream);

/* DDPM.reverse_q (ddpm.py:113-144): out = A_t x_t + B_t x0_pred + C_t z.  sigmoid_pred applies the "bce" branch
 * of DDPM.sample (ddpm.py:158-159) to x0_pred first. */
int ddpm_reverse_step(const float* xt, const float* x0_pred, const int64_t* t, const float* noise, const float* sched,
                      float* out, int N, int64_t chw, const uint64_t* seed, uint32_t stream_id, int sigmoid_pred,
                      ddpm_stream_t stream);

/* ELBOLoss.forward (losses/ELBO.py:36-76): *loss = mean(w_t * l(pred, target)), l = squared error (bce=0) or
 * BCE-with-logits (bce=1); weighted=0 is the use_elbo=False branch. */
int ddpm_elbo_fwd(const float* pred, const float* target, const int64_t* t, const float* sched, float* loss, int N,
                  int64_t chw, int bce, int weighted, ddpm_stream_t stream);
/* its gradient w.r.t. pred, times the upstream scalar *gscale (NULL = 1). */
int ddpm_elbo_bwd(const float* pred, const float* target, const int64_t* t, const float* sched, const float* gscale,
                  float* gpred, int N, int64_t chw, int bce, int weighted, ddpm_stream_t stream);

/* SinusoidalPositionEmbedBlock.forward on t / time_norm (constituent_blocks.py:261-279; unets.py:121,233).
 * out: bf16 [N][dim]. */
int ddpm_time_embedding(const int64_t* t, float time_norm, int dim, void* out, int N, ddpm_stream_t stream);

/* layout / precision bridges at the UNet boundary (no reference counterpart: the reference is NCHW fp32 throughout) */
int ddpm_nchw_f32_to_nhwc_bf16(const float* x, void* out, int N, int C, int HW, int cpad, ddpm_stream_t stream);
int ddpm_nhwc_bf16_to_nchw_f32(const void* x, float* out, int N, int C, int HW, int cpad, ddpm_stream_t stream);
/* Tanh backward of the UNet head (unets.py:183): dz = dy (1 - y^2), NCHW fp32 -> NHWC bf16 [cpad]. */
int ddpm_tanh_bwd_nchw_to_nhwc(const float* dy, const float* y, void* dz, int N, int C, int HW, int cpad, ddpm_stream_t stream);
/* *seed += odd constant: lets a replayed HIP graph draw fresh randomness every step. */
int ddpm_bump_seed(uint64_t* seed, ddpm_stream_t stream);

/* ---- dense convolutions on bf16 MFMA (conv_mfma.hip, conv_wgrad.hip) ---------------------------- */

typedef struct ddpm_conv_desc {
    int kind;                  /* DDPM_CONV_* / DDPM_DGRAD_* */
    int N, IH, IW;             /* dims of the tensor that is READ (input for forward kinds, dY for DGRAD kinds) */
    const void* in0;           /* NHWC bf16, C0 channels */
    const void* in1;           /* optional second source concatenated after in0 along C (torch.cat, constituent_blocks.py:231) */
    int C0, C1;
    const float* in_add;       /* optional fp32 [N][C0] broadcast-added to in0 before the conv (Z + temb, unets.py:128,237) */
    const uint8_t* in_mask;    /* optional dropout keep bits for in0, 1 byte per 8 elements (conv_block.py:58) */
    float in_mask_scale;       /* 1/(1-p) */
    const void* w;             /* bf16 [slices][CoutW][C0+C1], see ddpm_pack_* */
    const float* bias;         /* fp32 [Cout] or NULL */
    void* out;                 /* NHWC bf16 with channel stride out_cs, or NCHW fp32 (out_mode) */
    const void* res;           /* optional NHWC bf16 tensor added in the epilogue (same layout as out) */
    const uint8_t* out_mask;   /* optional dropout keep bits applied to the result before `res` is added */
    float out_mask_scale;
    int Cout, CoutW, out_cs;
    int act, out_mode;
} ddpm_conv_desc;

/* y = act(conv(x) + bias) [* mask] [+ res] for every kind above. */
int ddpm_conv_forward(const ddpm_conv_desc* d, ddpm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
