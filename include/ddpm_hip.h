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

/* activation codes (reference neural_nets/activations.py:102-114; the parameter-free ones; swish / hardswish carry a
 * learnable scalar and are not implemented) */
enum { DDPM_ACT_NONE = 0, DDPM_ACT_GELU = 1, DDPM_ACT_SILU = 2, DDPM_ACT_RELU = 3, DDPM_ACT_TANH = 4, DDPM_ACT_SIGMOID = 5,
       DDPM_ACT_HARDSIGMOID = 6, DDPM_ACT_RELU6 = 7, DDPM_ACT_LEAKY_RELU = 8 /* slope 0.05 */, DDPM_ACT_SELU = 9 };

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
/* Reproducible sums (no counterpart in the reference).  on != 0: the launchers whose kernels combine per-workgroup partial sums
 * with atomics (ddpm_elbo_fwd, ddpm_elbo_tanh_head, ddpm_colsum) run them as one workgroup, so two passes over the same numbers
 * give the same bits; the convolution entry points' sum_out stays an atomic sum (callers that need reproducible bits take the
 * sum with ddpm_spatial_sum instead).  Process-wide, default off (or DDPM_DETERMINISTIC=1); returns the previous setting. */
int ddpm_set_deterministic(int on);
int ddpm_deterministic(void);

/* ---- diffusion process (elementwise.hip) ------------------------------------------------------ */

/* Per-timestep table `sched`: fp32 [T+1][8], row t = { sqrt(abar_t), sqrt(1-abar_t), A_t, B_t, C_t, w_t, 0, 0 }
 * built by the host from the reference's schedule tensors (ddpm.py:44-62, ELBO.py:66-67). */

/* DDPM.q (ddpm.py:81-106): x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps.  x0: NCHW fp32.  noise: NCHW fp32, or
 * NULL to draw eps in-kernel from Philox keyed by *seed and stream_id.  Writes xt (NCHW fp32) and/or xt_nhwc
 * (NHWC bf16 with cpad channels, padding zeroed); either may be NULL.  H*W must be a multiple of 4. */
int ddpm_q_sample(const float* x0, const int64_t* t, const float* noise, const float* sched, float* xt, void* xt_nhwc,
                  int cpad, int N, int C, int HW, const uint64_t* seed, uint32_t stream_id, ddpm_stream_t stream);

/* The head of a training micro-step in one launch (base_trainer.py:202-204: torch.randint(1, T + 1, (N,)), DDPM.q; unets.py:121:
 * the sinusoidal embedding of t / time_norm): t_out[n] = the timestep of image n (t_in when given, else drawn as
 * ddpm_sample_timesteps with stream 3 does), x_t = q(x0, t) with in-kernel noise of stream 1 (or `noise`) written as the
 * channel-padded NHWC bf16 tensor the stem convolution reads (padding zeroed), sin_out = bf16 [N][t_dim] as
 * ddpm_time_embedding, and *seed advanced as ddpm_bump_seed does.  `done`: one zero-initialised uint32 of device memory the
 * launch uses to find its last workgroup (left at zero).  Same numbers as the separate calls. */
int ddpm_train_head(const float* x0, const int64_t* t_in, const float* noise, const float* sched, int64_t* t_out, void* xt_nhwc,
                    int cpad, void* sin_out, float time_norm, int t_dim, int N, int C, int HW, int T, uint64_t* seed, uint32_t* done,
                    ddpm_stream_t stream);
/* DDPM.reverse_q (ddpm.py:113-144): out = A_t x_t + B_t x0_pred + C_t z.  sigmoid_pred applies the "bce" branch
 * of DDPM.sample (ddpm.py:158-159) to x0_pred first.  z drawn in-kernel (noise == NULL) is the Philox sample of GLOBAL flat
 * element first_index + i (a multiple of 4): a rank that holds images [b, e) of a sharded sampling job passes
 * first_index = b * chw and draws exactly the numbers a single GPU would draw for those images.  out may be xt (in place). */
int ddpm_reverse_step(const float* xt, const float* x0_pred, const int64_t* t, const float* noise, const float* sched,
                      float* out, int N, int64_t chw, const uint64_t* seed, uint32_t stream_id, int sigmoid_pred,
                      uint64_t first_index, ddpm_stream_t stream);
/* out[i] = N(0,1) sample of global flat element first_index + i of stream stream_id of the (*seed) sequence: the start tensor
 * X_T of DDPM.sample (ddpm.py:152: randn), keyed like the reverse step's noise.  n and first_index multiples of 4. */
int ddpm_fill_normal(float* out, int64_t n, const uint64_t* seed, uint32_t stream_id, uint64_t first_index, ddpm_stream_t stream);

/* ELBOLoss.forward (losses/ELBO.py:36-76): *loss = mean(w_t * l(pred, target)), l = squared error (bce=0) or
 * BCE-with-logits (bce=1); weighted=0 is the use_elbo=False branch. */
int ddpm_elbo_fwd(const float* pred, const float* target, const int64_t* t, const float* sched, float* loss, int N,
                  int64_t chw, int bce, int weighted, ddpm_stream_t stream);
/* its gradient w.r.t. pred, times the upstream scalar *gscale (NULL = 1). */
int ddpm_elbo_bwd(const float* pred, const float* target, const int64_t* t, const float* sched, const float* gscale,
                  float* gpred, int N, int64_t chw, int bce, int weighted, ddpm_stream_t stream);
/* The squared-error ELBO of a tanh-headed denoiser (losses/ELBO.py:61-76 with reconstruction_method "mse", the Tanh of
 * unets.py:183 there) and the gradient chain down to the last convolution in one launch: *loss as ddpm_elbo_fwd, dz = the
 * channel-padded (16) NHWC bf16 gradient of the logits as ddpm_elbo_bwd + ddpm_tanh_bwd_nchw_to_nhwc give it (bit-identical),
 * bias_grad[c] += its per-channel sum as ddpm_colsum gives it.  pred = the network's output (after the tanh), NCHW fp32. */
int ddpm_elbo_tanh_head(const float* pred, const float* target, const int64_t* t, const float* sched, float* loss, void* dz,
                        float* bias_grad, int N, int C, int HW, int weighted, ddpm_stream_t stream);

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
/* Range check of the per-image timesteps (the reference's table look-ups raise IndexError for a timestep outside the
 * schedule, ddpm.py:46-66,159-201 there) that does not stall the host: safe[i] = t[i] clamped into [lo, hi], and an
 * out-of-range value sets flag[0] = 1, flag[1] = the value.  flag: two int32 in host-pinned (device-visible) memory. */
int ddpm_check_timesteps(const int64_t* t, int64_t* safe, int n, int lo, int hi, int32_t* flag, ddpm_stream_t stream);
/* Backward of the bare activation of a conv -> act unit without normalisation (conv_block.py:19-54 with norm = None):
 * dx = dy * act'(x) on NHWC bf16 [N][HW][C] (C = 8, 16, 32 or 64), and part[(n * P + p)][C] = column sums of dx over the
 * p-th of P equal pixel ranges of image n (partial rows of the convolution's bias gradient; P must divide HW). */
int ddpm_act_backward(const void* x, const void* dy, void* dx, float* part, int N, int HW, int C, int P, int act, ddpm_stream_t stream);
/* t[i] ~ U{1..T}, the per-image training timestep (base_trainer.py:202: torch.randint(1, T + 1, (N,))): Philox4x32-10
 * word i of the (*seed, stream_id) sequence mapped to the range by multiply-high.  t: device int64 [N]. */
int ddpm_sample_timesteps(int64_t* t, int N, int T, const uint64_t* seed, uint32_t stream_id, ddpm_stream_t stream);

/* Dataset transforms of the reference on the GPU (datasets/mnist.py:26-39, datasets/flowers102.py:28-38, augmentation/
 * augmentation.py:13): transforms.Resize (PIL bilinear, two fixed-point passes) -> ToTensor -> x*2-1 -> channel
 * replication (C0 = 1 -> C), optional horizontal flip per image (flip[n] != 0).  src: uint8 [N][H0][W0][C0]; dst: fp32
 * [N][C][H][W].  hcoef int32 [W][hk] / hbounds int32 [W][2] (first source column, count) and vcoef / vbounds likewise for
 * rows: PIL's 22-bit fixed-point resampling coefficients, computed by the host; NULL tables = that axis is not resized. */
int ddpm_prepare_images(const uint8_t* src, int N, int H0, int W0, int C0, float* dst, int C, int H, int W, const int32_t* hcoef,
                        const int32_t* hbounds, int hk, const int32_t* vcoef, const int32_t* vbounds, int vk, const uint8_t* flip,
                        ddpm_stream_t stream);

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
    const void* w;             /* bf16 [slices][w_rows][C0+C1]; rows [0, CoutW) of every slice are used (point w at a later row
                                * to compute a sub-range of output channels) */
    const float* bias;         /* fp32 [Cout] or NULL */
    void* out;                 /* NHWC bf16 with channel stride out_cs, or NCHW fp32 (out_mode).  May be NULL with a fused
                                * GroupNorm (gn_gamma .. gn_out below) at inference: the raw result is then not stored. */
    const void* res;           /* optional NHWC bf16 tensor added in the epilogue (same layout as out) */
    const uint8_t* out_mask;   /* optional dropout keep bits applied to the result before `res` is added */
    float out_mask_scale;
    int Cout, CoutW, out_cs;   /* real output channels written; rows computed (multiple of 32, >= Cout); channel stride of out */
    int w_rows;                /* rows per weight slice (>= CoutW) */
    int act, out_mode;
    float* sum_out;            /* optional fp32 [N][Cout] += per-image sum over output pixels of act(conv + bias), taken from the
                                * fp32 accumulators before mask / res: with a data-gradient kind this is the gradient of the
                                * broadcast time-embedding add (unets.py:128,237).  Caller zeroes it. */
    /* Optional fused GroupNorm + activation (+ residual) of the result (conv -> GroupNorm -> act, conv_block.py:19-54), where
     * ddpm_conv_gn_fusable() says one workgroup holds whole images and whole groups: `out` receives the raw convolution as
     * before (the backward pass needs it), gn_out [N][OH][OW][Cout] = act(GroupNorm(raw)) + gn_res, gn_stats [N][G][2] =
     * mean, rstd.  gn_gamma == NULL: no fused GroupNorm. */
    const float* gn_gamma; const float* gn_beta;
    void* gn_out; float* gn_stats; const void* gn_res;
    int gn_groups, gn_act;
    float gn_eps;
    void* out2; int act2;      /* optional second NHWC bf16 result act2(out) (out itself without activation): a conv -> act unit keeps
                                * the pre-activation for its backward pass and hands the activated tensor on, in one launch */
    /* Optional GroupNorm + activation BACKWARD applied to in0 on its way in (the data-gradient launch of a conv -> GroupNorm ->
     * act unit's own convolution): in0 = gradient of the unit's output, gnb_raw = the unit's raw
     * convolution result, gnb_stats [N][G][2] its statistics; the launch also writes gnb_draw (gradient of the raw result, which
     * the weight-gradient reads) and gnb_part [N][3][C] (per-image dgamma, dbeta, bias-gradient rows).  Only where
     * ddpm_conv_gnb_fusable() returns 1 (2x2 / 4x4 / 8x8 images: a tile holds whole images and all channels). */
    const void* gnb_raw; const float* gnb_gamma; const float* gnb_beta; const float* gnb_stats; float* gnb_part; void* gnb_draw;
    int gnb_groups, gnb_act;
    int C0_used;               /* 0, or a promise that only the first C0_used (1..4) channels of in0 are non-zero (an image stored
                                * zero-padded to 16 channels: the stem, and the head's data-gradient): lets a 3x3 convolution to 32
                                * channels run on a kernel that reads only those. */
    float* gn_part;            /* optional, large images: fp32 [N][gn_part_groups][T][2], T = ddpm_conv_gn_part_tiles(desc) > 0: every
                                * output tile stores the sum and the sum of squares of its (rounded) results per normalisation
                                * group, so that the GroupNorm that follows (ddpm_gn_desc.stat_part) needs no statistics pass of
                                * its own over the tensor */
    int gn_part_groups;
    const float* in_affine;    /* optional, inference on large images: in0 is the RAW input of a GroupNorm + activation
                                * (conv_block.py:89-96) that has not been applied: fp32 [N][C0][2] = (scale, shift) per image and
                                * channel from ddpm_gn_fold_affine; the loader computes act(x * scale + shift), rounded to bf16 as
                                * the GroupNorm launch would have stored it.  One input tensor, no in_add; in_mask applies after. */
    int in_affine_act;         /* DDPM_ACT none / gelu / silu */
    const float* gn_act_beta;  /* the fused GroupNorm's activation is swish / hardswish (neural_nets/activations.py:25-63): device
                                * pointer to its learnable scalar; forward only (the backward of such a unit runs on
                                * ddpm_gn_act_backward, which also produces the scalar's gradient) */
} ddpm_conv_desc;
int ddpm_conv_gn_fusable(const ddpm_conv_desc* d);
int ddpm_conv_gnb_fusable(const ddpm_conv_desc* d);
int ddpm_conv_gn_part_tiles(const ddpm_conv_desc* d);

/* y = act(conv(x) + bias) [* mask] [+ res] for every kind above. */
int ddpm_conv_forward(const ddpm_conv_desc* d, ddpm_stream_t stream);
/* Test hook (no counterpart in the reference): the general implicit-GEMM kernel picks its workgroup tile from the problem
 * size, so small test problems would never reach the tiles that large batches run on.  force > 0 (coded
 * wm * 100 + wn * 10 + nt: wm x wn waves, nt 32-channel column blocks per wave) makes the following ddpm_conv_forward calls
 * of the calling thread use that tile or fail; 0 restores the automatic choice; -1 changes nothing.  Returns the code of the
 * tile the thread's last general-kernel launch used (0 if none). */
int ddpm_conv_debug_tile(int force);
/* Which kernel the calling thread's last ddpm_conv_forward ran on (measurement and test hook, no reference counterpart):
 * the general tap-list kernel, the split-K kernel of 3x3 convolutions on 2x2 / 4x4 / 8x8 images, the vector-ALU kernel of the
 * image ends, or the whole-image kernel that finishes the unit's GroupNorm (or its backward) in the same launch. */
enum { DDPM_CONV_KERNEL_NONE = 0, DDPM_CONV_KERNEL_GENERAL = 1, DDPM_CONV_KERNEL_SMALL = 2, DDPM_CONV_KERNEL_THIN = 3, DDPM_CONV_KERNEL_IMAGE = 4,
       DDPM_CONV_KERNEL_PIPE = 5 /* the pipelined 256 x 128 tile kernel of deep, wide convolutions at large batch */,
       DDPM_CONV_KERNEL_PW_STREAM = 6 /* the streaming kernel of pointwise convolutions on large images at large batch */ };
int ddpm_conv_last_kernel(void);
/* Test / measurement hook (no counterpart in the reference): the pipelined kernel (3x3 stride 1 on 4x4 / 8x8 / 16x16 images and
 * 1x1, input channels a multiple of 64, output channels a multiple of 128, plain epilogue) is taken by problem size
 * (mode 0: at least one 256-position tile per CU), wherever the shape allows (1) or never (-1).  Per calling thread; returns
 * the previous mode. */
int ddpm_conv_pipe_mode(int mode);
/* Test / measurement hook (no counterpart in the reference): the streaming kernel of pointwise convolutions (1x1, one input
 * tensor of 32 / 64 / 128 channels, 32 / 64 / 128 output channels, at most 8192 weights, image area a multiple of 128, bias,
 * normalise-on-load and per-tile GroupNorm sums only) is taken by problem size (mode 0: at least 32 K positions), wherever the
 * shape allows (1) or never (-1).  Per calling thread; returns the previous mode. */
int ddpm_conv_pw_stream_mode(int mode);

/* ---- GroupNorm + activation (+ dropout mask, + residual) on NHWC bf16 (groupnorm.hip) ------------- */

typedef struct ddpm_gn_desc {
    const void* x;            /* NHWC bf16 [N][HW][C]: the tensor being normalised (a convolution's raw output) */
    const float* gamma;       /* fp32 [C] (GroupNorm.weight) */
    const float* beta;        /* fp32 [C] (GroupNorm.bias) */
    const void* res;          /* optional NHWC bf16 residual added last (conv_block.py:134-135) */
    const uint8_t* mask;      /* optional dropout keep bits applied to the activated output (the NEXT conv's nn.Dropout) */
    float mask_scale;
    void* y;                  /* forward: output; backward: dx */
    float* stats;             /* fp32 [N][G][2] = (mean, rstd): written by forward, read by backward */
    const void* dy;           /* backward: gradient w.r.t. the output (before `res` was added) */
    float* part;              /* backward: fp32 [N][3][C] per-image partial sums, every element written once:
                               * [n][0] = dgamma, [n][1] = dbeta, [n][2] = sum of dx over h,w (bias gradient of the producing
                               * conv).  Sum over n with ddpm_reduce_rows. */
    int N, HW, C, G;          /* G = 0: no normalisation, activation only */
    int act;
    float eps;
    float* scratch;           /* optional workspace of scratch_floats floats (ddpm_gn_scratch_floats): with it, images too large
                               * for one workgroup per (image, channel slab) are processed by many workgroups in two launches
                               * (partial sums, then apply); without it one workgroup walks the slab */
    int64_t scratch_floats;
    const float* act_beta;    /* act = swish / hardswish: device pointer to the activation's learnable scalar b
                               * (neural_nets/activations.py:40,60: z * sigmoid(b z), z * hardsigmoid(b z)); else unused */
    float* dbeta_part;        /* backward, swish / hardswish: fp32 [N][C], written once: per image and channel sums of
                               * dy * mask * d act / d b.  The scalar's gradient is the sum of all N*C entries
                               * (ddpm_reduce_rows entry with scalar = 1). */
    int fixed_stats;          /* 1: `stats` holds the statistics to normalise with (forward computes none; backward treats them
                               * as constants): BatchNorm2d in eval mode */
    const float* stat_part;   /* forward, optional: fp32 [N][G][stat_parts][2] partial sums (x, x^2) of the tensor, written by the
                               * convolution that produced it (ddpm_conv_desc.gn_part): where the many-workgroup path runs
                               * (ddpm_gn_scratch_floats > 0 and `scratch` given) its statistics launch is skipped */
    int stat_parts;
} ddpm_gn_desc;

/* y = [res +] mask * act(GroupNorm_G(x) * gamma + beta)     torch.nn.GroupNorm + activation sites:
 * constituent_blocks.py:68-73,83-88,169-174,182-187; conv_block.py:67-72,90-95,110-114; unets.py:80-85,162-167 */
int ddpm_gn_act_forward(const ddpm_gn_desc* d, ddpm_stream_t stream);
/* dx and the per-image partial sums of dgamma, dbeta, dbias of the above (no atomics: deterministic).  y (= dx) may be the
 * same tensor as dy, never the same as x. */
int ddpm_gn_act_backward(const ddpm_gn_desc* d, ddpm_stream_t stream);

/* GroupNorm statistics from the producer's partial sums, as what a consumer needs to normalise on load: stat_part = fp32
 * [N][G][parts][2] (ddpm_conv_desc.gn_part / ddpm_dwconv3x3_forward_stats) -> stats fp32 [N][G][2] (mean, rstd; optional) and
 * affine fp32 [N][C][2] = (gamma * rstd, beta - mean * gamma * rstd) for ddpm_conv_desc.in_affine. */
int ddpm_gn_fold_affine(const float* stat_part, int parts, const float* gamma, const float* beta, int N, int HW, int C, int G,
                        float eps, float* stats, float* affine, ddpm_stream_t stream);
/* floats of `scratch` the forward (backward = 0) or backward (1) call would use for this shape; 0 = none; -1 on error */
int64_t ddpm_gn_scratch_floats(const ddpm_gn_desc* d, int backward);
/* ddpm_gn_act_forward for callers that cannot allocate or pass a workspace (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arithmetic and results as
 * ddpm_gn_act_forward given a sufficient workspace and NO producer-side sums; d->scratch / d->scratch_floats and
 * d->stat_part / d->stat_parts are ignored (the statistics launch always runs: with stat_part the other entry point adds the
 * producer's sums in another order, so the last bits of its statistics may differ from this one's): the partial sums of the
 * many-workgroup path live in a device array built into the library, ddpm_gn_static_scratch_floats() floats (4 MiB) on every
 * GPU the library runs on, held until the process ends.  A shape that needs more (N * groups * pixel splits * 2 floats) is
 * rejected with an error.  The call allocates nothing.  All calls on one GPU share that array: calls that overlap on one GPU
 * (different streams) race; calls on one stream, or ordered by events, do not. */
int ddpm_gn_act_forward_static(const ddpm_gn_desc* d, ddpm_stream_t stream);
int64_t ddpm_gn_static_scratch_floats(void);

/* torch.nn.BatchNorm2d of the autoencoder blocks (neural_nets/autoencoders/constituent_blocks.py:67,85,160,178; base_ae.py:192).
 * Training mode IS ddpm_gn_act_forward / _backward on the tensor described as one image of N*H*W pixels with G = C groups
 * (statistics per channel over batch and space); stats then holds [C][2] = (batch mean, rstd).  These two calls complete the
 * module: the running-statistics update of a training step (momentum rule, unbiased variance, batch counter; num_batches may be
 * NULL), and the [C][2] statistics an eval-mode pass normalises with (pass them with fixed_stats = 1). */
int ddpm_bn_update_running(const float* stats, float* running_mean, float* running_var, int64_t* num_batches, int C, int64_t count,
                           float momentum, float eps, ddpm_stream_t stream);
int ddpm_bn_eval_stats(const float* running_mean, const float* running_var, float* stats, int C, float eps, ddpm_stream_t stream);

/* ---- depthwise 3x3 convolution, stride 1, padding 1, NHWC bf16 (dwconv.hip) ---------------------- */

/* nn.Conv2d(C, C, 3, padding=1, groups=C) of conv_block.py:79-88.  w: fp32 [9][C] (tap-major), bias fp32 [C] or NULL.
 * Also computes the data-gradient when given the tap-flipped weights and no bias. */
int ddpm_dwconv3x3_forward(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                           ddpm_stream_t stream);
/* The same convolution for large images, leaving the GroupNorm sums of conv_block.py:89 behind: part = fp32
 * [N][groups][T][2] partial sums (x, x^2) of the result, T = ddpm_dwconv3x3_stat_parts(...) > 0 (0: not available for the
 * shape), for ddpm_gn_desc.stat_part. */
int ddpm_dwconv3x3_stat_parts(int N, int H, int W, int C, int groups);
int ddpm_dwconv3x3_forward_stats(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                                 int groups, float* part, ddpm_stream_t stream);
/* Per-workgroup partial sums of the depthwise weight/bias gradient: part[b][t][c], t = 0..8 the taps of
 * dw (sum dy * shifted x), t = 9 the bias (sum dy); b < ddpm_dwconv3x3_wgrad_blocks(N,H,W,C).  Every element is written. */
/* Inference on large images: x is the RAW result of the expansion convolution, whose GroupNorm + activation (conv_block.py:67-72)
 * has not been applied; affine = fp32 [N][C][2] (scale, shift) from ddpm_gn_fold_affine, act = none / gelu / silu.  The kernel
 * normalises every input row once on its way into an LDS ring and runs the stencil from there: the GroupNorm launch and its
 * output tensor disappear.  Leaves the GroupNorm sums of its own result like ddpm_dwconv3x3_forward_stats (same layout).
 * Shapes: ddpm_dwconv3x3_affine_supported() (rows of 128 / 256 work items, H a multiple of 16, C <= 128). */
int ddpm_dwconv3x3_affine_supported(int N, int H, int W, int C, int groups);
int ddpm_dwconv3x3_forward_affine(const void* x, const float* affine, int act, const float* w, const float* bias, void* y, int N, int H,
                                  int W, int C, int groups, float* part, ddpm_stream_t stream);
int ddpm_dwconv3x3_wgrad(const void* x, const void* dy, float* part, int N, int H, int W, int C, ddpm_stream_t stream);
int ddpm_dwconv3x3_wgrad_blocks(int N, int H, int W, int C);

/* dst[e.dst_off + j] += sum_{r < e.rows} src[e.src_off + r * e.row_stride + j], j < e.width, for every entry: the second
 * stage of the partial-sum reductions above, all layers in one launch.  entries: device array. */
/* scalar = 1: dst[e.dst_off] += sum_{i < e.rows} src[e.src_off + i] (a contiguous run folded to one number by a whole
 * workgroup; row_stride and width are ignored): the gradient of an activation's learnable scalar. */
typedef struct ddpm_reduce_entry { int64_t src_off; int64_t dst_off; int32_t rows; int32_t row_stride; int32_t width; int32_t scalar; } ddpm_reduce_entry;
int ddpm_reduce_rows(const float* src, float* dst, const ddpm_reduce_entry* entries, int n_entries, int max_width,
                     ddpm_stream_t stream);

/* ---- weight gradient of the dense convolutions (conv_wgrad.hip) ---------------------------------- */

typedef struct ddpm_wgrad_desc {
    int kind;                 /* forward kind: DDPM_CONV_3X3_S1 / _S2 / DDPM_CONV_1X1 / DDPM_CONVT_2X2_S2 */
    int N, IH, IW;            /* dims of the forward INPUT tensor */
    const void* in0; const void* in1; int C0, C1;      /* forward inputs, exactly as given to ddpm_conv_forward */
    const float* in_add; const uint8_t* in_mask; float in_mask_scale;
    const void* dy;           /* NHWC bf16 gradient of the forward output, channel stride dy_cs */
    int Cout, dy_cs;
    float* dw;                /* fp32 [slices][Cout][C0+C1], accumulated (+=) with atomics; caller zeroes (unused when part is set) */
    float* part;              /* optional fp32 [ddpm_conv_wgrad_splits(d)][slices][Cout][C0+C1]: when set, every position split
                               * stores its partial sums there (each element written exactly once, no atomics) and the caller
                               * folds the rows into dw with ddpm_reduce_rows */
    int tiles_per_split;      /* 0: split the positions for a standalone launch (about 512 workgroups); > 0: at least this many
                               * 128-position tiles per split (grouped launches: fewer partial rows) */
} ddpm_wgrad_desc;

int ddpm_conv_wgrad(const ddpm_wgrad_desc* d, ddpm_stream_t stream);
/* Many weight-gradient problems in one launch per kernel variant (at most five launches).  A weight gradient is a leaf of
 * the backward pass (base_trainer.py:214 only needs them before optimizer.step), so a caller can defer all of a
 * network's to the end of backward.  plan() fills a host blob of ddpm_conv_wgrad_group_bytes(n) bytes from n descriptors;
 * the caller copies the blob to device memory (same bytes) and passes both to launch().  Descriptors, tensors and the
 * device blob must stay valid until the launches have run. */
int64_t ddpm_conv_wgrad_group_bytes(int n);
int ddpm_conv_wgrad_group_plan(const ddpm_wgrad_desc* descs, int n, void* host_blob, int64_t bytes);
int ddpm_conv_wgrad_group_launch(const void* host_blob, const void* dev_blob, ddpm_stream_t stream);
int ddpm_conv_wgrad_splits(const ddpm_wgrad_desc* d);   /* rows of `part` for this problem (shape fields only are read); -1 on error */
/* Rows of `part` when a standalone ddpm_conv_wgrad call (tiles_per_split = 0) takes this problem to one of the dedicated
 * kernels - the pipelined one for deep wide 3x3 stride-1 convolutions at large batch (conv_wgrad_pipe.hip) or the one for the
 * thin 3x3 convolutions at the image ends, 16 / 32 stored channels on 32 / 64 / 128-pixel-wide images (conv_wgrad_c32.hip) -,
 * 0 when it does not: a caller that groups its problems can launch these on their own.  (The reference leaves the choice of
 * kernel to torch: no counterpart.) */
int ddpm_conv_wgrad_pipe_splits(const ddpm_wgrad_desc* d);

/* db[C] += column sums of an NHWC bf16 tensor [rows][cs] (first C channels); fp32 atomics. */
int ddpm_colsum(const void* x, float* out, int64_t rows, int C, int cs, ddpm_stream_t stream);

/* out[n][c] (+)= sum over pixels of x[n][p][c]: gradient of the broadcast time-embedding add (unets.py:128,237). fp32 [N][C]. */
int ddpm_spatial_sum(const void* x, float* out, int N, int HW, int C, ddpm_stream_t stream);

/* Fill dropout keep-bit masks: for each of `nsites` sites, bytes[offset_i .. offset_i + nbytes_i) get 8 Bernoulli(keep)
 * bits each from Philox keyed by (*seed, site index).  offsets: device int64 [nsites+1] (prefix sums of bytes). */
int ddpm_dropout_masks(uint8_t* bytes, const int64_t* offsets, int nsites, int64_t total_bytes, float keep,
                       const uint64_t* seed, ddpm_stream_t stream);

/* ---- multi-tensor parameter plumbing (param_ops.hip) ------------------------------------------- */

/* One remapping job: dense A[s][r][k] (S x R x K; bf16 or fp32)  <->  fp32 B[r*a_r + k*a_k + off_s[s]] for r < Rv, k < Kv.
 * to_a = 1 fills A from B (zeros outside the valid range): used to derive the kernel-layout weight copies from the
 * torch-layout parameters (Conv2d OIHW, ConvTranspose2d IOHW, Linear out x in: base_trainer.py:118-130 keeps those as
 * the stored format).  to_a = 0 scatters A into B: kernel-layout gradients -> torch-layout .grad tensors. */
typedef struct ddpm_remap_entry {
    void* a; void* b;
    int32_t S, R, K, Rv, Kv;
    int32_t a_r, a_k;
    int32_t off_s[9];
    int32_t a_bf16;
    int32_t pad;
} ddpm_remap_entry;
/* chunks: device int32 pairs {entry index, first element of A / ddpm_remap_chunk_elems()} covering every entry. */
int ddpm_remap(const ddpm_remap_entry* entries /* device */, const int32_t* chunks /* device */, int n_chunks, int to_a,
               ddpm_stream_t stream);
int ddpm_remap_chunk_elems(void);

/* torch.optim.Adam step (optimizer/optimizer.py:3-8, base_trainer.py:229) over one flat fp32 buffer, optionally followed
 * by the EMA update of base_trainer.py:102-106,278-282 (ema may be NULL).  lr_dev (device float) overrides lr when given.
 * *step (device int64) is the number of steps taken so far and is incremented. */
int ddpm_adam_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* lr_dev, float lr,
                   float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale, int64_t* step,
                   ddpm_stream_t stream);
/* The same step on a range of the flat buffers (all pointers moved to the range's start, n = its length), for a driver that
 * updates a network's parameters in pieces as their gradients become final (the reference calls optimizer.step() once,
 * base_trainer.py:229: element-wise, so the pieces give the same numbers).  Every piece of one step reads the same *step;
 * only the call with bump != 0 (the step's last) increments it. */
int ddpm_adam_step_range(float* p, const float* g, float* m, float* v, float* ema, int64_t n, const float* lr_dev, float lr,
                         float beta1, float beta2, float eps, float weight_decay, float ema_decay, float grad_scale, int64_t* step,
                         int bump, ddpm_stream_t stream);

/* One SeparableInvertResidual block (conv_block.py:57-136: dropout -> 1x1 expansion -> GroupNorm -> act -> dropout ->
 * depthwise 3x3 -> GroupNorm -> act -> dropout -> 1x1 projection -> GroupNorm -> + input) as a single launch with the
 * block's activations resident in LDS.  x, y: NHWC bf16 [N][H][W][C].  we: bf16 [E][C], wp: bf16 [C][E] (the forward
 * weight-copy layout of ddpm_conv_forward), wd: fp32 [9][E]; biases and GroupNorm parameters fp32.  m_*: dropout keep
 * bits of the three sites (1 byte per 8 consecutive NHWC elements), all NULL outside training.  The eight saved-tensor
 * pointers are either all NULL (inference) or all given (training: raw convolution outputs, normalised activations
 * after dropout, GroupNorm statistics [N][groups][2], exactly what the unfused kernels would have produced).
 * Only the shapes for which ddpm_sir_supported() returns 1 are accepted. */
typedef struct ddpm_sir_desc {
    const void* x; void* y;
    const void* we; const float* be; const float* g1; const float* b1;
    const float* wd; const float* bd; const float* g2; const float* b2;
    const void* wp; const float* bp; const float* g3; const float* b3;
    const uint8_t* m_in; const uint8_t* m_e; const uint8_t* m_d;
    float mask_scale;
    float eps;
    void* raw_e; void* y_e; float* st1;
    void* raw_d; void* y_d; float* st2;
    void* raw_o; float* st3;
    int32_t N, C, H, W, E, groups, act, pad;
    const float* act_b1; const float* act_b2;   /* act = swish / hardswish: device pointers to the learnable scalars of the block's two
                                                 * activation modules (expansion.3, depthwise.3: conv_block.py:66-76,89-96) */
} ddpm_sir_desc;
int ddpm_sir_supported(int C, int H, int W, int E, int groups, int act);
int ddpm_sir_backward_supported(int C, int H, int W, int E, int groups, int act);   /* shapes ddpm_sir_backward accepts */
int ddpm_sir_forward(const ddpm_sir_desc* d, ddpm_stream_t stream);

/* Data-gradient chain of the same block (backward of ddpm_sir_forward) in one launch.  dy: gradient w.r.t. the block
 * output.  raw_e/raw_d/raw_o/y_e/st1-3: what ddpm_sir_forward saved.  wpd: bf16 [E][C], wed: bf16 [C][E] (the
 * data-gradient weight-copy layout of ddpm_conv_forward's DGRAD kinds for the projection / expansion), wdf: fp32 [9][E]
 * depthwise taps flipped.  Outputs: dx (gradient w.r.t. the block input, residual branch included), d_o [N][H][W][C] and
 * d_e [N][H][W][E] (gradients w.r.t. the raw projection / expansion outputs = the dY operands of ddpm_conv_wgrad), and
 * fp32 partial sums for ddpm_reduce_rows: part1/part2 [N][3][E], part3 [N][3][C] (rows: d gamma, d beta, column sum of the
 * raw gradient = bias gradient of the producing convolution; row 2 of part2 is not written), dwpart
 * [ddpm_sir_workgroups(H*W, N)][10][E] (nine depthwise tap gradients, then the depthwise bias gradient). */
typedef struct ddpm_sir_bwd_desc {
    const void* dy;
    const void* raw_e; const void* raw_d; const void* raw_o; const void* y_e;
    const float* st1; const float* st2; const float* st3;
    const float* g1; const float* b1; const float* g2; const float* b2; const float* g3;
    const void* wpd; const void* wed; const float* wdf;
    const uint8_t* m_in; const uint8_t* m_e; const uint8_t* m_d;
    void* dx; void* d_o; void* d_e;
    float* part1; float* part2; float* part3; float* dwpart;
    float mask_scale;
    int32_t N, C, H, W, E, groups, act;
    const float* act_b1; const float* act_b2;   /* act = swish / hardswish: the two activation modules' scalars, and ... */
    float* dbeta1; float* dbeta2;               /* ... fp32 [ddpm_sir_workgroups(H*W, N)][8] partial sums of their gradients (one per
                                                 * wave), for the scalar fold of ddpm_reduce_rows */
} ddpm_sir_bwd_desc;
int ddpm_sir_workgroups(int HW, int N);
int ddpm_sir_backward(const ddpm_sir_bwd_desc* d, ddpm_stream_t stream);

/* ---- the low-resolution stages of the UNet as ONE launch, activations resident in LDS (unet_lds.hip) --------------------
 * A workgroup carries `ipw` images through a program of layers: BaseUNetEncoder / BaseUNetDecoder.forward between two
 * points of the network (unets.py:127-131,230-237), i.e. DownBlock / UpBlock (constituent_blocks.py:135-147,227-243) and
 * SeparableInvertResidual (conv_block.py:132-136) bodies, layer by layer, without leaving the CU.  The program is a table
 * of operations on tiles of an LDS arena laid out by the caller: a tile = (byte offset, row stride in bytes), rows = the
 * pixels of the workgroup's images in image-major order, bf16 channels.  The arena's first ddpm_unet_lds_arena0() bytes
 * belong to the kernel.  Every pointer inside an operation is a DEVICE pointer.
 *   LOAD   p[0] = NHWC bf16 tensor (gstride channels per pixel, first channel gcol0) -> columns [col0, col0+cout) of dst
 *   STORE  columns [col0, col0+cout) of src -> p[0] (same addressing)
 *   CONV   kind = DDPM_CONV_3X3_S1 / _3X3_S2 / _1X1 / DDPM_CONVT_2X2_S2: src tile (w_in x w_in images, cin channels) ->
 *          columns [col0, col0+cout) of dst (w_out x w_out), raw result + bias rounded to bf16; p[0] = bf16 weights in the
 *          forward weight-copy layout [taps][gstride][cin], p[1] = fp32 bias or NULL; aux_off = ddpm_unet_lds_reduce_bytes()
 *          bytes of scratch when the layer has fewer than 8 output tiles of 32 x 32
 *   GN     in place on columns [col0, col0+cout) of dst: y = [aux tile +] mask * act(GroupNorm_groups(x) * p[0] + p[1]);
 *          act = 0 or the activation the launch was built for; p[2] = keep bits (1 per element, element index
 *          (image * pixels + pixel) * cout + channel) or NULL, scaled by mscale; aux_stride != 0: residual tile;
 *          training records: p[3] = raw input bf16 [N][pixels][cout], p[4] = result, p[5] = statistics fp32 [N][groups][2]
 *   DW     depthwise 3x3 + bias in place on dst: p[0] = fp32 [9][cout], p[1] = fp32 [cout]
 *   ADDT   columns [col0, col0+cout) of dst += p[0] (fp32 [N][cout]) broadcast over the pixels (unets.py:128,237)
 *   MASK   dst = src * keep bits of p[0] * mscale (the dropout on a block input, conv_block.py:57-58); p[0] = NULL: a copy
 * ddpm_unet_lds_check validates a HOST copy of the table and returns the LDS bytes the program needs (< 0: error);
 * ddpm_unet_lds_run validates the host copy, then launches ceil(N / ipw) workgroups that read the DEVICE copy. */
enum { DDPM_LOP_END = 0, DDPM_LOP_LOAD = 1, DDPM_LOP_STORE = 2, DDPM_LOP_CONV = 3, DDPM_LOP_GN = 4, DDPM_LOP_DW = 5, DDPM_LOP_ADDT = 6,
       DDPM_LOP_MASK = 7 };
typedef struct ddpm_lds_op {
    int32_t type, kind;
    int32_t src_off, src_stride, dst_off, dst_stride, aux_off, aux_stride;
    int32_t w_in, w_out, cin, cout, col0, groups, act, gstride, gcol0, pad0;
    float eps, mscale;
    const void* p[6];
} ddpm_lds_op;
int ddpm_unet_lds_arena0(void);
int ddpm_unet_lds_reduce_bytes(void);
int ddpm_unet_lds_max_bytes(void);
long long ddpm_unet_lds_check(const ddpm_lds_op* ops_host, int nops, int ipw);
int ddpm_unet_lds_run(const ddpm_lds_op* ops_dev, const ddpm_lds_op* ops_host, int nops, int N, int ipw, int act, ddpm_stream_t stream);

/* ---- all time-embedding MLPs of the network in one launch (embed_mlp.hip) --------------------------------------
 * EmbedBlock (constituent_blocks.py:282-328): out_i = Linear_1(act(Linear_0(sin))) for every stage i, sin = the
 * sinusoidal embedding [N][t_dim] bf16 (ddpm_time_embedding), t_dim = 32.  Width C[i], hidden width 2*C[i].
 * w0: bf16 [2C][t_dim], w1: bf16 [C][2C] (forward weight-copy layout), biases fp32, out: fp32 [N][C] (the operand of
 * ddpm_conv_desc.in_add).  raw0/h: optional saved tensors for training (bf16 [N][2C]: hidden pre-activation and
 * activation). */
#define DDPM_EMBED_MAX 8
typedef struct ddpm_embed_desc {
    const void* sin;
    int32_t N, t_dim, act, nmlp;
    int32_t C[DDPM_EMBED_MAX];
    const void* w0[DDPM_EMBED_MAX]; const float* b0[DDPM_EMBED_MAX];
    const void* w1[DDPM_EMBED_MAX]; const float* b1[DDPM_EMBED_MAX];
    float* out[DDPM_EMBED_MAX];
    void* raw0[DDPM_EMBED_MAX]; void* h[DDPM_EMBED_MAX];
} ddpm_embed_desc;
int ddpm_embed_forward(const ddpm_embed_desc* d, ddpm_stream_t stream);

/* Data path of the backward of all MLPs: dtemb fp32 [N][C] (gradient of out_i) -> dt16 (its bf16 copy, the dY operand of the
 * second layer's weight gradient) and dz bf16 [N][2C] (gradient of the hidden pre-activation, the dY operand of the first
 * layer's weight gradient); w1d: bf16 [2C][C] (data-gradient weight-copy layout).  db1 [ceil(N/32)][C] and db0
 * [ceil(N/32)][2C] receive per-row-tile bias-gradient partial sums for ddpm_reduce_rows. */
typedef struct ddpm_embed_bwd_desc {
    int32_t N, act, nmlp, pad;
    int32_t C[DDPM_EMBED_MAX];
    const float* dtemb[DDPM_EMBED_MAX]; const void* raw0[DDPM_EMBED_MAX]; const void* w1d[DDPM_EMBED_MAX];
    void* dt16[DDPM_EMBED_MAX]; void* dz[DDPM_EMBED_MAX];
    float* db1[DDPM_EMBED_MAX]; float* db0[DDPM_EMBED_MAX];
} ddpm_embed_bwd_desc;
int ddpm_embed_backward(const ddpm_embed_bwd_desc* d, ddpm_stream_t stream);

/* ---- attention core and per-sample norm of ConditionalModule (attention.hip) -------------------------------------
 * nn.MultiheadAttention's scaled-dot-product core (constituent_blocks.py:357-368,426: 8 heads, batch_first, dropout on
 * the attention probabilities): o = dropout(softmax(q k^T / sqrt(dh))) v per (sample, head).  The in/out projections
 * are 1x1 convolutions (ddpm_conv_forward).  Element (n, token, head, d) of q is q[(n*S + token)*ldq + head*dh + d]
 * (bf16), likewise k, v (Sk tokens), o, and the gradients; a packed [N][S][3C] projection output is passed as three
 * pointers with ld = 3C.  dh is 4 or a multiple of 8 up to 64; row strides are multiples of 8 (4 when dh = 4) and the
 * base pointers aligned likewise.  lse: fp32 [N][heads][S], written by forward when not NULL (base-2 log-sum-exp of the
 * scaled scores) and read by backward.  drop_p > 0 draws the keep mask from Philox keyed by (*seed, stream_id); backward
 * must be given the same seed value and stream_id.  The keep probability is rounded to a multiple of 1/256. */
typedef struct ddpm_attn_desc {
    const void* q; const void* k; const void* v;
    int32_t ldq, ldk, ldv;
    void* o;                   /* forward: output; backward: the forward output (input) */
    int32_t ldo;
    float* lse;
    const void* dout;          /* backward: gradient w.r.t. o, row stride ldd */
    int32_t ldd;
    void* dq; void* dk; void* dv;
    int32_t lddq, lddk, lddv;
    int32_t N, S, Sk, heads, dh;
    float drop_p;
    const uint64_t* seed;
    uint32_t stream_id;
} ddpm_attn_desc;
int ddpm_attention_forward(const ddpm_attn_desc* d, ddpm_stream_t stream);
int ddpm_attention_backward(const ddpm_attn_desc* d, ddpm_stream_t stream);

/* y = GroupNorm(num_groups=1, C)(act(a) + b) * gamma + beta over all S*C elements of each sample
 * (constituent_blocks.py:425-428: a = attention output projection, b = module input, act = identity; :431-434: a = Linear
 * output, b = previous norm output, act = the block activation).  a, b, y: bf16 [N][S][C]; b may be NULL.
 * stats: fp32 [N][2] (mean, rstd).  Backward: du = gradient w.r.t. (act(a) + b) (= gradient of b), da = du * act'(a)
 * (may be NULL when act is DDPM_ACT_NONE), part fp32 [N][3][C] = per-sample d gamma, d beta and column sums of da (the bias
 * gradient of the Linear that produced a), every element written once.  C = 8 x a power of two, at most 512. */
typedef struct ddpm_sample_norm_desc {
    const void* a; const void* b;
    const float* gamma; const float* beta;
    void* y;
    float* stats;
    const void* dy; void* du; void* da;
    float* part;
    int32_t N, S, C, act;
    float eps;
    const float* act_beta;   /* act = swish / hardswish: the activation's learnable scalar (device pointer) */
    float* dbeta_part;       /* backward, swish / hardswish: fp32 [N], per-sample partial sums of the scalar's gradient */
} ddpm_sample_norm_desc;
int ddpm_sample_norm_forward(const ddpm_sample_norm_desc* d, ddpm_stream_t stream);
int ddpm_sample_norm_backward(const ddpm_sample_norm_desc* d, ddpm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
