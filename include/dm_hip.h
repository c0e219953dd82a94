/*
 * dm_hip.h -- C ABI of the MI355X-native sampling path (libdm_hip.so).
 *
 * The reference (lbarseghyan/diffusion-models) has no FFI or plugin interface: its
 * boundary is a Python method surface on nn.Module subclasses.  Each entry point
 * below names the reference method it stands behind (paths relative to the
 * reference checkout; DD = denoising-diffusion-pytorch/denoising_diffusion,
 * LD = latent-diffusion/ldm).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from dm_last_error() (thread-local).  Nothing throws across the ABI.
 *   - `const float*` tensor arguments are DEVICE pointers, contiguous fp32, NCHW at
 *     this boundary (the reference's layout), unless the name says `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - a handle is bound to one device and is not thread-safe.
 *   - PyTorch (or any caller) owns inputs and outputs; the library owns only its
 *     repacked weights and a workspace that grows outside graph capture.
 */
#ifndef DM_HIP_H
#define DM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_MAX_STAGES 8

/* text_mode */
#define DM_TEXT_NONE 0
#define DM_TEXT_CONCAT 1 /* DD/denoising_diffusion_text_conditional.py:108-115,146-152 */
#define DM_TEXT_CROSS 2  /* DD/denoising_diffusion_text_conditional.py:120-125,173-198 */

/* Constructor arguments of the reference Unet that fix shapes
 * (DD/denoising_diffusion.py:234-252; text variant
 * DD/denoising_diffusion_text_conditional.py:97). */
typedef struct dm_unet_cfg {
    int32_t dim;
    int32_t init_dim;       /* 0 = dim */
    int32_t out_dim;        /* 0 = channels */
    int32_t channels;       /* image / latent channels the model predicts */
    int32_t input_channels; /* channels init_conv sees (self-cond / image-cond widen it) */
    int32_t n_stages;
    int32_t dim_mults[DM_MAX_STAGES];
    int32_t full_attn[DM_MAX_STAGES];
    int32_t attn_heads;
    int32_t attn_dim_head;
    int32_t text_mode;
    int32_t text_emb_dim;
    float sinusoidal_theta;
    int32_t learned_sinusoidal_dim; /* 0: SinusoidalPosEmb(dim, theta); > 0: RandomOrLearnedSinusoidalPosEmb of that
                                       dimension (DD/denoising_diffusion.py:86-101, parameter time_mlp.0.weights) --
                                       dm_unet_forward only: DenoisingDiffusion refuses such a U-Net (:456-457) */
    int32_t attn_heads_stage[DM_MAX_STAGES]; /* Unet(attn_heads = (h0, h1, ..)): heads of stage i's attention (:294, :310,
                                                :327; mid_attn takes the last stage's, :324); 0 = attn_heads */
} dm_unet_cfg;

typedef struct dm_unet dm_unet;

const char* dm_last_error(void);
/* ABI version of this header; bump on any signature change. */
int dm_abi_version(void);

/* ---- U-Net (replaces DD/denoising_diffusion.py:233-390 `Unet`) -------------------- */

int dm_unet_create(const dm_unet_cfg* cfg, int device, dm_unet** out);
void dm_unet_destroy(dm_unet* u);

/* One call per state_dict() entry of the reference Unet (names without the
 * `model.` prefix, e.g. "downs.0.0.block1.proj.weight"); data is a HOST pointer
 * to contiguous fp32 in the reference's own layout (OIHW conv weights, [out,in]
 * linear weights).  Shapes are checked against the configuration. */
int dm_unet_set_param(dm_unet* u, const char* name, const float* data_host, const int64_t* shape, int ndim);
/* number of parameters still missing (0 = complete) */
int dm_unet_missing_params(dm_unet* u);
/* read back the handle's host copy of one parameter (n = its element count): `Unet.state_dict()` / `.parameters()` of a
 * handle that is not in training mode (a training handle reads its device-resident state: dm_unet_get_param) */
int dm_unet_get_param_host(dm_unet* u, const char* name, float* out_host, int64_t n);
/* repack weights into kernel layouts and upload; must follow the last set_param */
int dm_unet_finalize(dm_unet* u);

/* In-place weight refresh for the caller of record: Trainer.train samples from `self.ema.ema_model` after every EMA
 * update (DD/denoising_diffusion.py:1190,1198,1216), i.e. the same architecture with new values each time.
 * dm_unet_update_param replaces the host copy of one parameter of a FINALIZED handle (same name / shape rules as
 * dm_unet_set_param; unchanged values are detected and cost nothing); dm_unet_refresh then re-packs the layers whose
 * parameters changed into the SAME device buffers (no reallocation: workspace, captured step graph and every device
 * pointer stay valid).  The call synchronises the device. */
int dm_unet_update_param(dm_unet* u, const char* name, const float* data_host, const int64_t* shape, int ndim);
int dm_unet_refresh(dm_unet* u);
/* how many times a denoise-step graph has been captured on this handle (diagnostics / tests: one per shape) */
int dm_unet_graph_captures(dm_unet* u);
/* bytes of the activation workspace the handle currently owns (grown by the largest call so far; activations are
 * released to it as soon as their last consumer is enqueued, so a B=256 step works in a few hundred MB) */
int64_t dm_unet_workspace_bytes(dm_unet* u);

/* Unet.forward(x, time, x_self_cond=None) DD/denoising_diffusion.py:349-390 and the
 * text variant forward(x, time, text_emb) DD/denoising_diffusion_text_conditional.py:131-214.
 *   x      (B, input_channels, H, W)   time (B,) int64 device   ctx (B, ctx_tokens, text_emb_dim) or NULL
 *   out    (B, out_dim, H, W)
 * H and W must be divisible by 2^(n_stages-1) (assert at :350). */
int dm_unet_forward(dm_unet* u, const float* x, const int64_t* time, const float* ctx, int ctx_tokens,
                    float* out, int B, int H, int W, void* stream);

/* ---- samplers (replace DenoisingDiffusion.p_sample_loop / ddim_sample,
 *      DD/denoising_diffusion.py:647-664 and :666-708) ------------------------------------
 *
 * The per-step scalar coefficients are computed by the HOST exactly as the reference
 * computes them (fp32 tensor arithmetic on the schedule buffers) and passed in:
 *   DDPM step i (t = times[i]):   c[0]=sqrt_recip_alphas_cumprod[t]  c[1]=sqrt_recipm1_alphas_cumprod[t]
 *                                 c[2]=posterior_mean_coef1[t]       c[3]=posterior_mean_coef2[t]
 *                                 c[4]=exp(0.5*posterior_log_variance_clipped[t])   c[5]= (t>0) ? 1 : 0
 *   DDIM step i (t, t_next):      c[0], c[1] as above  c[2]=sqrt(alpha_next)  c[3]=c  c[4]=sigma
 *                                 c[5]= (t_next<0) ? 0 : 1   (0: img = x_start, :686-689)
 * coefs_host has n_steps rows of DM_COEFS floats.
 *
 *   x_T        (B,C,H,W) start noise (draw #0 of the reference)
 *   noise      NULL -> device Philox noise from `seed`; else (n_steps, B,C,H,W) injected noise,
 *              row i used by step i (rows of steps that take no noise are ignored)
 *   sample_offset  index of this call's first sample in the GLOBAL batch (0 for an unsharded call).  Philox
 *              counters are the global element index, (sample_offset*C*H*W + e) / 4, so a batch sharded over ranks
 *               draws exactly the noise the unsharded batch draws: cat(shards) == whole, bit for bit.
 *              Generate x_T with dm_randn(..., draw 0, element_offset = sample_offset*C*H*W) for the same property.
 *   out        (B,C,H,W); (x+1)/2 applied when unnormalize != 0 (:663,:707)
 *   all_steps  NULL, or (n_steps+1, B,C,H,W) receiving x_T and every iterate
 *              (return_all_timesteps; the caller permutes to (B, n_steps+1, ...))
 *   use_graph  replay one denoise step as a hipGraph n_steps times.  The instantiated graph is cached on the handle
 *              and reused by later calls of the same (kind, B, H, W, ctx_tokens, cond_channels); seed, offset, step
 *              tables, x_T, ctx and cond are device data or copied into handle-owned buffers, not captured arguments.
 */
#define DM_COEFS 8
#define DM_SAMPLER_DDPM 0
#define DM_SAMPLER_DDIM 1
/* what the U-Net output means (`objective` of DenoisingDiffusion.__init__, DD/denoising_diffusion.py:443; branches of
 * model_predictions :607-624).  pred_v reads c[6]=sqrt_alphas_cumprod[t], c[7]=sqrt_one_minus_alphas_cumprod[t]
 * (predict_start_from_v :588-592) from the step table. */
#define DM_OBJ_PRED_NOISE 0
#define DM_OBJ_PRED_X0 1
#define DM_OBJ_PRED_V 2

int dm_sample(dm_unet* u, int kind, int n_steps, const int64_t* times_host, const float* coefs_host,
              const float* x_T, const float* noise, uint64_t seed, uint64_t sample_offset, const float* ctx,
              int ctx_tokens, float* out, float* all_steps, int B, int H, int W, int unnormalize, int use_graph,
              void* stream);

/* The same loop for the image-conditional variant (replaces ImageConditionalDenoisingDiffusion.p_sample_loop /
 * ddim_sample, DD/denoising_diffusion_image_conditional.py:156-224; its Unet.forward concatenates `cond` behind x in
 * front of init_conv, :51-55).  cond is (B, cond_channels, H, W) fp32 on the device, constant over the loop; the
 * handle must have input_channels == channels + cond_channels. */
int dm_sample_cond(dm_unet* u, int kind, int n_steps, const int64_t* times_host, const float* coefs_host,
                   const float* x_T, const float* noise, uint64_t seed, uint64_t sample_offset, const float* ctx,
                   int ctx_tokens, const float* cond, int cond_channels, float* out, float* all_steps, int B, int H,
                   int W, int unnormalize, int use_graph, void* stream);

/* The general form of the two calls above: every option of the loop in one struct (zero-initialise it; unused fields
 * stay 0 / NULL).  objective: DM_OBJ_*.  self_condition != 0: the handle was built with input_channels == 2*channels and
 * every step feeds the U-Net [x_start of the previous step | x] (zeros at the first step), as p_sample_loop / ddim_sample
 * do when model.self_condition is set (DD/denoising_diffusion.py:352-354,:657,:683); not combined with cond. */
typedef struct dm_sample_args {
    int32_t kind;           /* DM_SAMPLER_* */
    int32_t objective;      /* DM_OBJ_* */
    int32_t self_condition;
    int32_t n_steps;
    const int64_t* times_host;
    const float* coefs_host;
    const float* x_T;
    const float* noise;
    uint64_t seed;
    uint64_t sample_offset;
    const float* ctx;
    int32_t ctx_tokens;
    int32_t cond_channels;
    const float* cond;
    float* out;
    float* all_steps;
    int32_t B, H, W;
    int32_t unnormalize;
    int32_t use_graph;
    int32_t reserved_;
    void* stream;
} dm_sample_args;
int dm_sample_ex(dm_unet* u, const dm_sample_args* args);

/* N(0,1) noise from the library's Philox4x32-10 stream (what dm_sample uses when noise == NULL);
 * element e of the tensor of draw `draw` uses counter ((element_offset + e)/4, draw) under key `seed`
 * (element_offset % 4 == 0; draw 0 = x_T, draw i+1 = the noise of loop step i). */
int dm_randn(float* out, int64_t n, uint64_t seed, uint64_t draw, uint64_t element_offset, void* stream);

/* ---- VAE decode (replaces VQModel.decode, LD/models/autoencoder.py:113-116 ->
 *      Decoder.forward LD/modules/diffusionmodules/model.py:552-585) ----------------------- */
typedef struct dm_decoder_cfg {
    int32_t ch;
    int32_t out_ch;
    int32_t n_levels;
    int32_t ch_mult[DM_MAX_STAGES];
    int32_t num_res_blocks;
    int32_t n_attn_res;
    int32_t attn_resolutions[DM_MAX_STAGES];
    int32_t resolution;
    int32_t z_channels;
    int32_t embed_dim;
} dm_decoder_cfg;

typedef struct dm_decoder dm_decoder;
int dm_decoder_create(const dm_decoder_cfg* cfg, int device, dm_decoder** out);
void dm_decoder_destroy(dm_decoder* d);
int dm_decoder_set_param(dm_decoder* d, const char* name, const float* data_host, const int64_t* shape, int ndim);
int dm_decoder_missing_params(dm_decoder* d);
int dm_decoder_finalize(dm_decoder* d);
/* z (B, embed_dim, h, w) -> out (B, out_ch, h*2^(n_levels-1), w*2^(n_levels-1)) */
int dm_decoder_forward(dm_decoder* d, const float* z, float* out, int B, int h, int w, void* stream);

/* ---- VAE encode (replaces VQModel.encode, LD/models/autoencoder.py:102-106 -> Encoder.forward
 *      LD/modules/diffusionmodules/model.py:451-476 -> quant_conv -> VectorQuantizer2 of taming-transformers):
 *      the condition image of ImageConditionalLatentDiffusion (LD/models/latent_diffusion_image_conditional.py:55-66)
 * Parameter names: VQModel.state_dict() entries "encoder.*", "quant_conv.*", "quantize.embedding.weight". ---- */
typedef struct dm_encoder_cfg {
    int32_t ch;
    int32_t in_channels;
    int32_t n_levels;
    int32_t ch_mult[DM_MAX_STAGES];
    int32_t num_res_blocks;
    int32_t n_attn_res;
    int32_t attn_resolutions[DM_MAX_STAGES];
    int32_t resolution;
    int32_t z_channels;
    int32_t embed_dim;
    int32_t n_embed;
    int32_t double_z;
} dm_encoder_cfg;
typedef struct dm_encoder dm_encoder;
int dm_encoder_create(const dm_encoder_cfg* cfg, int device, dm_encoder** out);
void dm_encoder_destroy(dm_encoder* e);
int dm_encoder_set_param(dm_encoder* e, const char* name, const float* data_host, const int64_t* shape, int ndim);
int dm_encoder_missing_params(dm_encoder* e);
int dm_encoder_finalize(dm_encoder* e);
/* x (B, in_channels, H, W) device fp32 -> zq (B, embed_dim, H/f, W/f) = z + (nearest code - z); optional outputs:
 * pre_quant (same shape, the quant_conv output = VQModel.encode_to_prequant) and indices (B*h*w int32 code ids).
 * zq may be NULL when only pre_quant is wanted. */
int dm_encoder_forward(dm_encoder* e, const float* x, float* zq, float* pre_quant, int32_t* indices, int B, int H, int W,
                       void* stream);

/* ---- single operators (the kernels behind the calls above, exposed so that parity tests
 *      can check each one against the reference module it replaces) -------------------------
 * All tensors NCHW fp32 device pointers; weights in the reference layout, DEVICE pointers. */

/* nn.Conv2d forward (stride 1) with optional fused extras used by the U-Net:
 *   in = cat(in0, in1) along C (in1 may be NULL);  up2: nearest x2 before the conv
 *   (DD/denoising_diffusion.py:48-52);  residual (B,Cout,Ho,Wo) added to the result or NULL. */
int dm_op_conv2d(const float* in0, int C0, const float* in1, int C1, const float* weight, const float* bias,
                 const float* residual, float* out, int B, int H, int W, int Cout, int ksize, int pad, int up2,
                 void* stream);
/* Downsample: pixel-unshuffle(2) + conv1x1 (DD/denoising_diffusion.py:54-58); weight (Cout, 4*C, 1, 1) */
int dm_op_downsample(const float* in, int C, const float* weight, const float* bias, float* out, int B, int H,
                     int W, int Cout, void* stream);
/* RMSNorm.forward (DD/denoising_diffusion.py:66-67) */
int dm_op_rmsnorm(const float* x, const float* g, float* out, int B, int C, int H, int W, void* stream);
/* Block.forward (DD/denoising_diffusion.py:113-122); scale/shift (B,Cout) or NULL */
int dm_op_block(const float* x, int Cin, const float* weight, const float* bias, const float* g,
                const float* scale, const float* shift, float* out, int B, int H, int W, int Cout, void* stream);
/* LinearAttention.forward (DD/denoising_diffusion.py:173-193) */
int dm_op_linear_attention(const float* x, const float* norm_g, const float* mem_kv, const float* w_qkv,
                           const float* w_out, const float* b_out, const float* out_g, float* out, int B, int C,
                           int H, int W, int heads, int dim_head, void* stream);
/* Attention.forward (DD/denoising_diffusion.py:215-229, DD/attend.py:109-124) */
int dm_op_attention(const float* x, const float* norm_g, const float* mem_kv, const float* w_qkv,
                    const float* w_out, const float* b_out, float* out, int B, int C, int H, int W, int heads,
                    int dim_head, void* stream);
/* one DDPM / DDIM update on (n) elements given eps = model output (meaning per `objective`, DM_OBJ_*); c = DM_COEFS
 * floats (host); x_start (optional) receives the clamped x_0 estimate of the step (pred_x_start of model_predictions) */
int dm_op_sampler_update(int kind, int objective, const float* x, const float* eps, const float* noise,
                         const float* c_host, float* out, float* x_start, int64_t n, void* stream);

/* ---- sample consumer: the InceptionV3 feature extractor behind the reference's FID and
 *      Inception-score evaluators (DD/fid_evaluation.py:41-51 -> pytorch_fid.inception.InceptionV3;
 *      DD/inception_score_evaluation.py:70-92 -> torchvision.models.inception_v3).  The layer graph is host code
 *      (diffusion-models_amd/inception.py); these are its operators.  Activations are NHWC fp32 DEVICE pointers.
 *      Both libraries and their pretrained weights are absent here: parity unpinned, checked against the oracle's
 *      restatement of the published architectures. -------------------------------------------------------------- */
typedef struct dm_conv dm_conv;
/* nn.Conv2d(Cin, Cout, (KH, KW), stride, (pad_h, pad_w)) with an optional ReLU; weight_host OIHW, bias_host or NULL
 * (BasicConv2d = conv + BatchNorm(eval) + ReLU is passed with the BatchNorm folded into weight and bias) */
int dm_conv_create(const float* weight_host, const float* bias_host, int Cout, int Cin, int KH, int KW, int stride,
                   int pad_h, int pad_w, int relu, int device, dm_conv** out);
void dm_conv_destroy(dm_conv* c);
/* in: (B, H, W, Cin) NHWC, or (B, Cin, H, W) when in_nchw != 0; out: (B, Ho, Wo, Cout) NHWC.  All dm_conv handles of a
 * device share one scratch workspace (K-split partial sums): their calls must be ordered on one stream. */
int dm_conv_forward(dm_conv* c, const float* in, int in_nchw, int B, int H, int W, float* out_nhwc, void* stream);
/* F.max_pool2d / F.avg_pool2d on NHWC: mode 0 max, 1 avg with count_include_pad=True, 2 avg with count_include_pad=False */
int dm_op_pool2d(const float* in, float* out, int B, int H, int W, int C, int k, int stride, int pad, int mode,
                 void* stream);
/* F.interpolate(x, (Ho, Wo), mode="bilinear", align_corners=False) of an NCHW batch, written NHWC, then
 * scale[c] * v + shift[c] (scale / shift: C floats on the device) */
int dm_op_resize_bilinear(const float* in_nchw, float* out_nhwc, int B, int C, int H, int W, int Ho, int Wo,
                          const float* scale_dev, const float* shift_dev, void* stream);
/* torch.cat along channels, one source at a time: dst[row][c_off + c] = src[row][c] */
int dm_op_copy_channels_nhwc(const float* src, int Cs, float* dst, int Cd, int c_off, int64_t rows, void* stream);
/* adaptive_avg_pool2d(x, (1, 1)) on NHWC: out (B, C) */
int dm_op_global_avgpool(const float* in_nhwc, float* out, int B, int HW, int C, void* stream);
/* nn.Linear: y (R, O) = x (R, I) W^T + b; weight (O, I) and bias on the device */
int dm_op_linear(const float* x, const float* weight, const float* bias, float* y, int R, int I, int O, void* stream);

/* ---- training step: DenoisingDiffusion.forward / p_losses, DD/denoising_diffusion.py:805-900, and
 *      the backward pass of the same U-Net (what `accelerator.backward(loss)` computes in Trainer.train, :1162-1176).
 *      Unconditional U-Net (no text / image condition, no self-conditioning), dropout 0, loss_weight from the caller.
 *      Gradients are kept on the handle in the reference's parameter layouts (state_dict() shapes). ------------------- */

/* allocate the gradient buffers and pack the input-gradient convolutions (the forward kernels run on 180-degree rotated,
 * Cin<->Cout transposed weights); dm_unet_refresh re-packs them together with the forward weights afterwards */
int dm_unet_train_enable(dm_unet* u);
/* total floats of the flat gradient buffer (parameters in state-dict order, each padded to a multiple of 4), or -1 */
int64_t dm_unet_grad_floats(dm_unet* u);
/* the flat gradient buffer itself (device pointer, dm_unet_grad_floats floats): data-parallel training (accelerate / DDP in
 * the reference's Trainer) all-reduces it in place -- ONE collective for all gradients -- before dm_unet_optimizer_step */
int dm_unet_grads_flat(dm_unet* u, float** ptr_out, int64_t* n_out);
/* Gradient buckets for data-parallel training -- what torch DDP's 25 MB buckets are to the reference's Trainer under accelerate
 * (DD/denoising_diffusion.py:971-974, :1175): the flat buffer is laid out bucket by bucket in the order the backward pass
 * completes them (convolution weights of the stages it leaves first; the last bucket holds the last stages and every
 * parameter only the end of the pass completes), DM_TRAIN_BUCKET_MB (default 25) per bucket.
 *   dm_unet_train_buckets(u, enable): returns the number of buckets; with enable != 0 every later dm_unet_loss_backward runs a
 *     bucket's weight gradients as soon as the pass has left the bucket's stages, and records an event.
 *   dm_unet_train_bucket(u, i, &off, &n, wait, stream): bucket i's span of the flat buffer in floats; with wait != 0 `stream`
 *     is made to wait for the bucket's event of the last pass, so that a collective enqueued on it runs beside the rest of
 *     the pass (the caller joins the streams before dm_unet_optimizer_step). */
int dm_unet_train_buckets(dm_unet* u, int enable);
int dm_unet_train_bucket(dm_unet* u, int i, int64_t* off_out, int64_t* n_out, int wait, void* wait_stream);
/* copy the gradient of one parameter (names as in dm_unet_set_param) into a DEVICE buffer of the parameter's size */
int dm_unet_get_grad(dm_unet* u, const char* name, float* out_dev, void* stream);
/* One p_losses call (:823-889) + backward:
 *   x = q_sample(x_start, t, noise) (:813-821);  out = Unet(x, t);  target per `objective` (DM_OBJ_*, :864-872);
 *   loss = loss_scale * mean_b( loss_weight[t_b] * mean((out - target)^2) ) (:874-878, :889);  every parameter gradient.
 * x_start, noise: (B, C, H, W) device, x_start already normalised to [-1, 1];  t_host: (B) timesteps;
 * coef_host: (B, DM_TRAIN_COEFS = 8) = sqrt_alphas_cumprod[t_b], sqrt_one_minus_alphas_cumprod[t_b], loss_weight[t_b], 0,
 * sqrt_recip_alphas_cumprod[t_b], sqrt_recipm1_alphas_cumprod[t_b], 0, 0 -- the values `extract` gathers (:394-397).
 * self_cond (Unet(self_condition=True), :846-855): 0 off; 1 the U-Net sees [0 | x]; 2 it sees [x_start | x] with x_start
 * predicted (unclipped, without gradient) by a first forward pass on [0 | x] -- the caller flips the reference's coin.  loss_scale = 1 / gradient_accumulate_every and accumulate != 0 adds the gradients to
 * what the buffers hold (the micro-batch loop of Trainer.train, :1164-1176).  loss_out_host receives the scalar loss;
 * model_out (optional, device) the U-Net output.  cond (optional): the condition image (B, cond_channels, H, W) of the
 * image-conditional variant, concatenated behind x in front of init_conv (DD/denoising_diffusion_image_conditional.py:51-55,
 * p_losses :251-311).  ctx (optional): the text embeddings (B, ctx_tokens, text_emb_dim) of the text-conditional variant
 * (DD/denoising_diffusion_text_conditional.py:131-214, p_losses :476-542), concat or cross-attention per the handle's
 * text_mode.  noise_q (optional): the noise q_sample mixes in when it is not `noise` itself -- with immiscible=True the
 * reference's q_sample re-assigns the noise rows inside (:815-817) while p_losses keeps the unpermuted tensor as the
 * target (:865).  The call synchronises the stream unless loss_out_host is NULL (then see dm_unet_train_scalar). */
int dm_unet_loss_backward(dm_unet* u, const float* x_start, const int64_t* t_host, const float* coef_host,
                          const float* noise, const float* noise_q, const float* cond, int cond_channels, const float* ctx,
                          int ctx_tokens,
                          int self_cond, int objective, float loss_scale, int accumulate, float* loss_out_host,
                          float* model_out, int B, int H, int W, void* stream);
/* The same call with its arguments in one struct, plus the hybrid (KL) term of p_losses (:880-897):
 *   loss_terms: 1 the weighted MSE (what dm_unet_loss_backward computes), 2 the KL term alone, 3 both in one pass.  The
 *     reference evaluates the KL term through p_mean_variance, i.e. a SECOND forward pass of the U-Net with gradients: without
 *     dropout that pass repeats the first one bit for bit and loss_terms = 3 is the same loss and gradient; with dropout the
 *     second pass draws new masks -- the caller then runs loss_terms = 1 followed by loss_terms = 2 with accumulate = 1.
 *   coef_stride: floats per row of coef_host, 8 (rows as above) or 12: [3] = 1 if t_b > 0 else 0 (the reference's mask),
 *     [8] posterior_mean_coef1[t_b], [9] posterior_mean_coef2[t_b], [10] posterior_variance[t_b],
 *     [11] posterior_log_variance_clipped[t_b] -- required by the KL term.
 *   kl_scale = 0.001 / (mask.sum() + 1e-8), formed by the caller in fp32 as the reference does (:893-895).
 * The KL term divides by posterior_variance[t], which is 0 at t = 0, before the mask multiplies (inf * 0): a batch holding a
 * t = 0 sample has a NaN loss and NaN gradients in the reference and here. */
typedef struct dm_train_args {
    const float* x_start;
    const int64_t* t_host;
    const float* coef_host;
    int coef_stride;
    const float* noise;
    const float* noise_q;
    const float* cond;
    int cond_channels;
    const float* ctx;
    int ctx_tokens;
    int self_cond;
    int objective;
    float loss_scale;
    int accumulate;
    float* loss_out_host;
    float* model_out;
    int B, H, W;
    void* stream;
    int loss_terms;
    float kl_scale;
} dm_train_args;
int dm_unet_loss_backward_ex(dm_unet* u, const dm_train_args* a);
/* The rest of one Trainer.train iteration (:1178-1190) on device-resident state: the master parameters, the Adam moments and
 * the EMA copy live in flat device buffers in the reference layouts; after the update every packed weight buffer the
 * kernels read is rebuilt on the device (pack_kernels.hip, bit-identical to the host packers).
 *   dm_unet_optimizer_step: clip_grad_norm_(max_grad_norm; <= 0: off), Adam(lr, (beta1, beta2), eps) step; the total
 *                           gradient norm (before clipping) goes to grad_norm_out_host when given.
 *   dm_unet_ema_update:     copy != 0: ema <- online;  else ema <- ema * decay + online * (1 - decay)
 *   dm_unet_get_param:      one tensor of the online parameters (which = 0), the EMA copy (1) or Adam's exp_avg (2) /
 *                           exp_avg_sq (3) into a device buffer -- what Trainer.save (:1100-1113) writes as 'model' / 'ema' /
 *                           'opt'
 *   dm_unet_set_train_tensor: the inverse for which = 1, 2, 3 (Trainer.load, :1115-1133; the online parameters go through
 *                           dm_unet_set_param + dm_unet_refresh);  dm_unet_adam_step: torch.optim.Adam's `step` counter,
 *                           returned (set first when set_to >= 0; -1: not a training handle)
 *   dm_unet_train_sync:     device -> host copies + dm_unet_refresh, after which the handle samples with the trained weights
 *                           (the sampling entry points refuse to run on stale fused packs until then)
 *   dm_unet_check_device_pack: self-check, number of packed buffers whose device packer differs from the host packer */
int dm_unet_optimizer_step(dm_unet* u, float lr, float beta1, float beta2, float eps, float max_grad_norm,
                           float* grad_norm_out_host, void* stream);
/* The loop's scalars without a host round trip (the reference's loss is a device tensor until Trainer calls loss.item(),
 * DD/denoising_diffusion.py:1173): with loss_out_host == NULL dm_unet_loss_backward, and with grad_norm_out_host == NULL
 * dm_unet_optimizer_step, return as soon as their kernels are enqueued; dm_unet_train_scalar copies the loss of the last
 * loss / backward call (which = 0) or the total gradient norm of the last optimiser step (which = 1) to a DEVICE float. */
int dm_unet_train_scalar(dm_unet* u, int which, float* out_dev, void* stream);
int dm_unet_ema_update(dm_unet* u, float decay, int copy, void* stream);
int dm_unet_get_param(dm_unet* u, const char* name, int which, float* out_dev, void* stream);
int dm_unet_set_train_tensor(dm_unet* u, const char* name, int which, const float* src_dev, void* stream);
long long dm_unet_adam_step(dm_unet* u, long long set_to);
int dm_unet_train_sync(dm_unet* u);
int dm_unet_check_device_pack(dm_unet* u);
/* nn.Dropout(p) of the Blocks in training mode (Unet(dropout = p), :111,121; the shipped ddpm_cifar.yaml trains with 0.1).
 * Masks are Philox4x32-10 draws keyed by (seed, call, block index): every dm_unet_loss_backward call after this one draws
 * fresh masks; the backward pass re-creates the masks of its forward from the key.  The VALUES differ from torch's
 * generator, the semantics (Bernoulli(1 - p) keep, scale 1 / (1 - p), after the activation, before the residual add) do not.
 * dm_op_dropout_mask writes the factor (0 or 1 / (1 - p)) the block_index-th Block (forward order) of call number `call`
 * applies, n elements in (B, H, W, C) order -- parity tests hand these masks to the oracle. */
int dm_unet_train_dropout(dm_unet* u, float p, uint64_t seed);
int dm_op_dropout_mask(float* out, int64_t n, float p, uint64_t seed, uint64_t call, int block_index, void* stream);
/* q_sample (:813-821) on its own: out = coef[b][0] * x_start + coef[b][1] * noise, coef_host (B, 12): rows as in dm_train_args */
int dm_op_q_sample(const float* x_start, const float* noise, const float* coef_host, float* out, int B, int per_sample,
                   void* stream);
/* The elementwise helpers of DenoisingDiffusion as callable methods -- predict_start_from_noise, predict_noise_from_start,
 * predict_v, predict_start_from_v, the posterior mean of q_posterior (DD/denoising_diffusion.py:570-601), and through them
 * model_predictions (:603-626) / p_mean_variance (:628-636) with per-sample timesteps: coef_host (B, 2) = the two values
 * `extract` gathers;  mode 0: out = c0 * x + c1 * y;  mode 1: out = (c0 * x - y) / c1;  clamp != 0: to [-1, 1] afterwards.
 * dm_op_mask_mix: out = a * mask + b * (1 - mask), the guide step of ddim_sample_guided (:754); all (n) device floats. */
int dm_op_lincomb(const float* x, const float* y, const float* coef_host, float* out, int B, int64_t per_sample, int mode,
                  int clamp, void* stream);
int dm_op_mask_mix(const float* a, const float* b, const float* mask, float* out, int64_t n, void* stream);
/* offset noise (:830-834): noise[b][c][:] += strength * offset[b][c]  (noise (B, C, H, W) in place, offset (B, C)) */
int dm_op_offset_noise(float* noise, const float* offset, float strength, int BC, int HW, void* stream);
/* immiscible diffusion's noise assignment (:805-817): out[i][j] = || x[i] - y[j] ||_2 over D floats (torch.cdist of the
 * flattened batches; the caller runs scipy's linear_sum_assignment on it, as the reference does, on the host) and the row
 * gather dst[i] = src[idx[i]] (idx: n host indices) */
int dm_op_cdist(const float* x, const float* y, float* out, int n, int m, int64_t D, void* stream);
int dm_op_gather_rows(const float* src, const int64_t* idx_host, float* dst, int n, int64_t D, void* stream);

/* Backward of the single operators above (what autograd computes for the reference module), for parity tests of each
 * piece of the training step.  All tensors NCHW fp32 device pointers; gradient outputs have the shape of the tensor they
 * differentiate; optional outputs may be NULL. */
/* nn.Conv2d backward: 3x3 / pad 1 (up2: the conv ran on the nearest-x2 upsampled cat(in0, in1)) or 1x1; dy (B,Cout,Ho,Wo) */
int dm_op_conv2d_bwd(const float* in0, int C0, const float* in1, int C1, const float* weight, const float* dy, float* d_in0,
                     float* d_in1, float* d_weight, float* d_bias, int B, int H, int W, int Cout, int ksize, int pad, int up2,
                     void* stream);
/* Downsample backward (DD/denoising_diffusion.py:54-58) */
int dm_op_downsample_bwd(const float* in, int C, const float* weight, const float* dy, float* d_in, float* d_weight,
                         float* d_bias, int B, int H, int W, int Cout, void* stream);
/* Block backward (:113-122); d_scale / d_shift (B, Cout) */
int dm_op_block_bwd(const float* x, int Cin, const float* weight, const float* bias, const float* g, const float* scale,
                    const float* shift, const float* dy, float* dx, float* d_weight, float* d_bias, float* d_g,
                    float* d_scale, float* d_shift, int B, int H, int W, int Cout, void* stream);
/* RMSNorm backward (:66-67) */
/* nn.Linear backward on its own (time_mlp, ResnetBlock.mlp; DD/denoising_diffusion.py:127-130, :280-285): dx (R, I) = dy W,
 * d_weight (O, I) (+)= dy^T x, d_bias (O) (+)= column sums of dy; any of the three outputs may be NULL */
int dm_op_linear_bwd(const float* x, const float* weight, const float* dy, float* dx, float* d_weight, float* d_bias, int R,
                     int I, int O, int accumulate, void* stream);
int dm_op_rmsnorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* dg, int B, int C, int H, int W,
                      void* stream);
/* LinearAttention / Attention backward (:173-193, :215-229) */
int dm_op_linear_attention_bwd(const float* x, const float* norm_g, const float* mem_kv, const float* w_qkv,
                               const float* w_out, const float* b_out, const float* out_g, const float* dy, float* dx,
                               float* d_norm_g, float* d_mem_kv, float* d_w_qkv, float* d_w_out, float* d_b_out,
                               float* d_out_g, int B, int C, int H, int W, int heads, int dim_head, void* stream);
int dm_op_attention_bwd(const float* x, const float* norm_g, const float* mem_kv, const float* w_qkv, const float* w_out,
                        const float* b_out, const float* dy, float* dx, float* d_norm_g, float* d_mem_kv, float* d_w_qkv,
                        float* d_w_out, float* d_b_out, int B, int C, int H, int W, int heads, int dim_head, void* stream);

/* ---- measurement (bench.py's roofline leg; not part of the reference surface) -------------
 * While enabled, every convolution / fused-attention launch is bracketed by two HIP events recorded on
 * the stream the kernel is launched on.  Do not combine with use_graph.  dm_profile_enable(1) first
 * measures the interval of an EMPTY-kernel bracket (dispatch + event packets); dm_profile_read
 * synchronises the device, subtracts that interval from every bracket (total_ms is kernel execution
 * time), aggregates the recorded launches per kernel instance, and clears the records.
 * total_flops / total_bytes are ALGORITHMIC (2*k*k*Cin*Cout*pixels; input + output + weights
 * each moved once), see DESIGN.md. */
typedef struct dm_profile_row {
    char kernel[64];
    int64_t launches;
    double total_ms;
    double total_flops;
    double total_bytes;
} dm_profile_row;
int dm_profile_enable(int on);
int dm_profile_read(dm_profile_row* rows, int max_rows, int* n_rows);

#ifdef __cplusplus
}
#endif
#endif /* DM_HIP_H */
