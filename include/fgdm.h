/* libfgdm_hip.so -- C ABI of the MI355X-native FG-DM sampling engine.
 *
 * Drop-in boundary for the hot path named in BASELINE.json.  Each entry point states the
 * reference interface it replaces (paths relative to the DeepakSridhar/fgdm checkout).  Conventions:
 *   - return 0 on success, negative FGDM_ERR_* otherwise; nothing throws across the ABI;
 *     fgdm_last_error() gives a message for the last failing call on that engine;
 *   - all tensor arguments are DEVICE pointers owned by the caller (e.g. torch tensor.data_ptr()), except
 *     fgdm_load_tensor which accepts host or device memory;
 *   - latents / eps are fp32 NCHW [B,4,H,W]; timesteps int64 [B]; context fp32 [B,77,ctx_dim];
 *     hints fp32 NCHW [B,3,8H,8W] in [0,1];
 *   - the engine owns weights and workspace; one engine per device; not thread-safe; all work is enqueued on
 *     the stream passed in (a hipStream_t cast to void*; NULL = default stream), no implicit synchronisation.
 */
#ifndef FGDM_H
#define FGDM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FGDM_MAX_LEVELS 8
#define FGDM_MAX_CONTROLNETS 4

typedef struct fgdm_engine fgdm_engine;

/* Hyper-parameters of UNetModel / ControlNet: the keys of models/config.yaml:33-48 and
 * controlnet/models/cldm_v15_canny.yaml:21-53 (use_spatial_transformer=True, transformer_depth=1, legacy=False). */
typedef struct fgdm_config {
    int32_t in_channels, out_channels, model_channels, num_res_blocks;
    int32_t n_levels;
    int32_t channel_mult[FGDM_MAX_LEVELS];
    int32_t n_attention_resolutions;
    int32_t attention_resolutions[FGDM_MAX_LEVELS];
    int32_t num_heads, context_dim;
    int32_t use_adapter;       /* FG-DM self-prompt adapter (openaimodel.py:551-556): 0 none (no_prompting), 1 Adapter,
                                * 2 TimeAdapter (use_time_adapter=True; ldm/modules/encoders/adapter.py:387-417) */
    int32_t n_controlnets;     /* 0..FGDM_MAX_CONTROLNETS ControlNet twin encoders (cldm.py:545-790) */
    int32_t hint_channels;     /* 3 */
    int64_t workspace_bytes;   /* initial activation slab; 0 = default; grows on demand */
    /* First-stage decoder (AutoencoderKL ddconfig, models/config.yaml:55-69); vae_ch = 0: engine without a decoder */
    int32_t vae_ch, vae_n_levels;
    int32_t vae_ch_mult[FGDM_MAX_LEVELS];
    int32_t vae_num_res_blocks, vae_z_channels, vae_out_ch;
    /* Text encoder (transformers.CLIPTextModel behind FrozenCLIPEmbedder, ldm/modules/encoders/modules.py:137-162);
     * clip_layers = 0: engine without one.  openai/clip-vit-large-patch14: 12 layers, width 768, 12 heads, mlp 3072,
     * vocab 49408, 77 positions. */
    int32_t clip_layers, clip_width, clip_heads, clip_mlp, clip_vocab, clip_max_len;
    int32_t n_extra_adapters;  /* AdaptUNetModel num_prompts - 1 (openaimodel.py:993-999): further Adapters whose features are
                                * summed onto the FG-DM adapter's; needs use_adapter = 1 */
} fgdm_config;

#define FGDM_DTYPE_F32 0
#define FGDM_DTYPE_F16 1

/* flags for fgdm_apply_model */
#define FGDM_FLAG_USE_ORIGINAL 1      /* UNetModel.forward_original: skip the adapter (openaimodel.py:818-822) */
#define FGDM_FLAG_ONLY_MID_CONTROL 2  /* ControlledUnetModel only_mid_control (cldm.py:43-44) */
#define FGDM_FLAG_NO_CONTROL 4        /* cond['c_concat'] is None branch (cldm.py:842-843) */
#define FGDM_FLAG_CFG_PAIRS 8         /* the batch is cat([x]*2), cat([t]*2) of a classifier-free-guidance step (ddim.py:222-
                                       * 226; same pcond, and the cached hint covers B/2 rows): rows b and b + B/2 differ only
                                       * in the context, so the network up to its first cross-attention runs once on B/2 rows.
                                       * Results are bit-identical to the call without the flag.  The flag is the CALLER's
                                       * assertion that the halves are equal: with it only rows [0, B/2) of x, t and pcond are
                                       * read for that prefix (pcond must still hold B rows); the host mirrors verify pcond
                                       * before setting it (fgdm_amd/models.py). */

int fgdm_create(const fgdm_config* cfg, int device, fgdm_engine** out);
void fgdm_destroy(fgdm_engine* e);
const char* fgdm_last_error(const fgdm_engine* e);   /* e == NULL: why the last fgdm_create failed */

/* Parameter table: the reference state_dict keys the engine expects.  UNet keys carry the prefix
 * "model.diffusion_model." (adapter: "model.diffusion_model.adapter."), ControlNet k uses "control_model."
 * (k = 0) or "control_model_<k>." -- the names load_state_dict sees at
 * scripts/txt2img_fgdm_inference.py:23-38 and controlnet/cldm/model.py:12-28.  Works without a GPU. */
int fgdm_param_count(const fgdm_config* cfg);
int fgdm_param_info(const fgdm_config* cfg, int index, char* name, int name_cap, int64_t* shape, int* ndim);

/* Replaces load_state_dict: copy one tensor (host or device memory, contiguous) into the engine's staging. */
int fgdm_load_tensor(fgdm_engine* e, const char* key, const void* data, int dtype, const int64_t* shape, int ndim);
/* Repack all loaded tensors into the kernel layouts (fp16, K-contiguous, GEGLU-interleaved ...) in HBM. */
int fgdm_finalize_weights(fgdm_engine* e);

/* ControlNet.input_hint_block (cldm.py:655-671,796): t-independent, so computed once per image batch and cached
 * inside the engine for ControlNet `cn`; the reference recomputes it on every apply_model call. */
int fgdm_set_hint(fgdm_engine* e, int cn, const float* hint, int B, int Hh, int Wh, void* stream);

/* LatentDiffusion.apply_model / DiffusionWrapper.forward -> UNetModel.forward (ldm/models/diffusion/ddpm.py:1035-1136,
 * 1829-1848; openaimodel.py:808-884) and, when the engine has ControlNets and FGDM_FLAG_NO_CONTROL is not set,
 * ControlLDM.apply_model (controlnet/cldm/cldm.py:836-849) using the hints cached by fgdm_set_hint:
 * eps = UNet(x, t, ctx, control = sum_k scales_k * ControlNet_k(x, hint_k, t, ctx)).
 * control_scales: n_controlnets * 13 floats (cldm.py:823) or NULL for 1.0.  pcond: optional adapter input
 * (openaimodel.py:838-841) fp32 NCHW or NULL (= x).  Timesteps: int64 `t`, or fractional fp32 `t_float` when non-NULL
 * (timestep_embedding accepts fractional t, util.py:165; DPM-Solver feeds (t_continuous - 1/N) * 1000).
 * ctx: fp32 [B,77,context_dim], or NULL to use the context registered with fgdm_set_context. */
/* AdaptUNetModel.forward(..., conds=[...]) (openaimodel.py:1263-1320): n_conds <= n_extra_adapters condition latents,
 * each fp32 NCHW [B,4,H,W] (device).  adapters[k](conds[k]) does not depend on x or t, so the summed features are
 * computed here once and added in every later fgdm_apply_model of the same B, H, W (the reference's `control` prompt is
 * fgdm_apply_model's `pcond`).  n_conds = 0 clears them (conds=None). */
int fgdm_set_adapter_conds(fgdm_engine* e, const float* const* conds, int n_conds, int B, int H, int W, void* stream);

/* The conditioning is loop-invariant (ddim.py:147-162 hands the same `cond` to every step): fgdm_set_context projects
 * ctx fp32 [B,77,context_dim] (device) through every cross-attention layer's to_k / to_v (attention.py:183-186) once;
 * fgdm_apply_model calls with ctx == NULL and the same B then reuse those projections.  A later fgdm_set_context or
 * fgdm_finalize_weights replaces / drops them. */
int fgdm_set_context(fgdm_engine* e, const float* ctx, int B, void* stream);
int fgdm_apply_model(fgdm_engine* e, const float* x, const int64_t* t, const float* t_float, const float* ctx,
                     const float* pcond, const float* control_scales, int B, int H, int W, int flags, float* eps_out,
                     void* stream);

/* Text conditioning: FrozenCLIPEmbedder.forward after tokenisation (ldm/modules/encoders/modules.py:153-158):
 * ids int64 [B, T] (device; T <= clip_max_len, the tokenizer's padded max_length) -> last_hidden_state fp32 [B, T, W]
 * (device).  Needs clip_layers > 0 and the cond_stage_model.transformer.text_model.* tensors loaded. */
int fgdm_clip_encode(fgdm_engine* e, const int64_t* ids, int B, int T, float* out, void* stream);

/* First-stage decode: LatentDiffusion.decode_first_stage (ldm/models/diffusion/ddpm.py:832-889, plain branch) =
 * AutoencoderKL.decode(scale * z) (ldm/models/autoencoder.py:330-333; Decoder.forward,
 * ldm/modules/diffusionmodules/model.py:532-560).  z fp32 NCHW [B,4,H,W] (device), scale = 1 / scale_factor;
 * image fp32 NCHW [B, vae_out_ch, f*H, f*W], f = 2^(vae_n_levels-1).  Needs an engine created with vae_ch > 0 and the
 * first_stage_model.decoder.* / first_stage_model.post_quant_conv.* tensors loaded. */
int fgdm_vae_decode(fgdm_engine* e, const float* z, int B, int H, int W, float scale, float* image, void* stream);

/* Stage boundary of the two-factor chain (device pointers; byte work, bit-exact to the reference's expressions):
 *  fgdm_image_to_uint8: fp32 NCHW image -> uint8 NHWC.  mode 0 = uint8(255 * clamp((x+1)/2, 0, 1))
 *      (scripts/txt2img_fgdm_inference.py:245,249-252); mode 1 = uint8(clip(x*127.5+127.5, 0, 255))
 *      (controlnet/initialize_cn.py:101).
 *  fgdm_resize_linear_uint8: cv2.resize(img, (Wo, Ho), interpolation=cv2.INTER_LINEAR) on uint8 NHWC
 *      (scripts/txt2img_fgdm_inference.py:258; OpenCV generic fixed-point path, see oracle/boundary.py).
 *  fgdm_uint8_to_hint: control = float(img) / 255, NHWC -> fp32 NCHW (controlnet/initialize_cn.py:78-80). */
int fgdm_image_to_uint8(const float* image, int B, int C, int H, int W, int mode, uint8_t* out, void* stream);
int fgdm_resize_linear_uint8(const uint8_t* src, int B, int H, int W, int C, int Ho, int Wo, uint8_t* dst, void* stream);
int fgdm_uint8_to_hint(const uint8_t* src, int B, int H, int W, int C, float* hint, void* stream);

/* ControlNet.forward alone (cldm.py:792-813): the 13 residual tensors as fp32 NCHW, written back-to-back into
 * `out` in the order the reference returns them.  Test/inspection entry; apply_model never materialises them. */
int fgdm_controlnet(fgdm_engine* e, int cn, const float* x, const int64_t* t, const float* ctx, int B, int H, int W,
                    float* out, int64_t out_capacity_floats, void* stream);

/* One block of the loaded graph, addressed by its state-dict prefix -- the parity-test entry for the reference's block-level
 * modules: a TimestepEmbedSequential ("model.diffusion_model.input_blocks.4.", "...middle_block.", "...output_blocks.0.",
 * "control_model.input_blocks.1."; openaimodel.py:75-90) made of ResBlock (openaimodel.py:275-301), SpatialTransformer
 * (ldm/modules/attention.py:275-292), Downsample / Upsample (openaimodel.py:114-180), or the FG-DM Adapter
 * ("model.diffusion_model.adapter."; ldm/modules/encoders/adapter.py:334-346).
 * x fp32 NCHW [B,C,H,W]; x_skip: for decoder blocks the second half of th.cat([h, hs.pop()], 1) (openaimodel.py:869) fp32 NCHW
 * [B,Cs,H,W], else NULL; emb fp32 [B, 4*model_channels]: the `emb` the reference hands to the block (NULL if it has no
 * ResBlock); ctx fp32 [B,77,context_dim] (NULL if it has no SpatialTransformer).  out: fp32 NCHW (Adapter: its four
 * feature maps back to back); *out_numel = floats written. */
int fgdm_run_block(fgdm_engine* e, const char* prefix, const float* x, int C, const float* x_skip, int Cs, const float* emb,
                   const float* ctx, int B, int H, int W, float* out, int64_t out_capacity_floats, int64_t* out_numel,
                   void* stream);

/* One fused sampler update on fp32 tensors of n elements.
 * fgdm_ddim_step: p_sample_ddim's CFG combine + x_prev/pred_x0 (ldm/models/diffusion/ddim.py:243,254-268;
 * controlnet/cldm/ddim_hacked.py:192,203-231).  e_uncond NULL = no CFG; noise NULL = sigma term skipped;
 * any of x_prev / pred_x0 / e_out may be NULL. */
int fgdm_ddim_step(const float* x, const float* e_cond, const float* e_uncond, float cfg_scale, float a_t,
                   float a_prev, float sigma_t, float sqrt_one_minus_at, const float* noise, float* x_prev,
                   float* pred_x0, float* e_out, int64_t n, void* stream);
/* Adams-Bashforth eps combination of p_sample_plms (ldm/models/diffusion/plms.py:224-232); order 1..3. */
int fgdm_plms_combine(const float* e_t, const float* e1, const float* e2, const float* e3, int order,
                      float* e_prime, int64_t n, void* stream);
/* y = ca*a + cb*b (b may be NULL): the Heun-like first PLMS step (plms.py:219-223) and mask blends (ddim.py:151-154) */
int fgdm_axpby(const float* a, float ca, const float* b, float cb, float* y, int64_t n, void* stream);
/* y = a*mask + (1-mask)*b, mask already expanded to n elements: the inpainting blend of ddim.py:151-154 /
 * ldm/models/diffusion/ddpm.py:1419-1421. */
int fgdm_mask_blend(const float* a, const float* b, const float* mask, float* y, int64_t n, void* stream);
/* p_sample / p_mean_variance / q_posterior (ldm/models/diffusion/ddpm.py:284-297,1260-1323) for one timestep. */
int fgdm_ancestral_step(const float* x, const float* eps, float sqrt_recip_ac, float sqrt_recipm1_ac, float coef1,
                        float coef2, float std, const float* noise, float* out, int64_t n, void* stream);

/* Whole DDIM loop on the device (DDIMSampler.sample with eta = 0 and no host callbacks, ddim.py:58-177;
 * ddim_hacked.py:55-178).  alphas/alphas_prev/sqrt_one_minus_alphas: S floats (host) in ddim_timesteps order,
 * timesteps: S int64 (host).  cond/uncond ctx fp32 [B,77,ctx]; uncond NULL or cfg_scale == 1 disables CFG.
 * CFG runs as ONE 2B batch (cat([uncond, cond]), ddim.py:222-226); x is updated in place. */
int fgdm_sample_ddim(fgdm_engine* e, float* x, const float* cond, const float* uncond, float cfg_scale, int S,
                     const int64_t* timesteps, const float* alphas, const float* alphas_prev,
                     const float* sqrt_one_minus_alphas, const float* control_scales, int B, int H, int W,
                     int flags, void* stream);

/* Built-in kernel timer (no reference counterpart: the reference only prints wall-clock, scripts/txt2img.py:381-396).
 * Between begin and end every stride-th kernel launch is bracketed by HIP events on the launch stream
 * (work / bytes / time totals then refer to the bracketed launches only).
 * out: 4 classes x {device ms, launches, algorithmic work, algorithmic HBM bytes (igemm only)}; classes: 0 implicit GEMM (flops), 1 attention (flops),
 * 2 GroupNorm+LayerNorm (bytes), 3 im2col (bytes).  fgdm_profile_end synchronises the device. */
int fgdm_profile_begin(fgdm_engine* e, int stride /* bracket every stride-th launch; 1 = all */);
int fgdm_profile_end(fgdm_engine* e, double* out);
/* Activation workspace: peak bytes in use during the last calls, and bytes reserved from HBM. */
int fgdm_workspace_stats(fgdm_engine* e, int64_t* peak_bytes, int64_t* reserved_bytes);
/* Grouped twin launches (the UNet encoder and the ControlNets of ControlLDM.apply_model, controlnet/cldm/cldm.py:836-849, share every
 * layer shape): launches replayed from recorded walks since fgdm_create, how many of them were fused launches, and how many
 * problems those carried.  Diagnostic: the tests assert that the fused kernel really ran. */
int fgdm_launch_stats(fgdm_engine* e, int64_t* replayed_launches, int64_t* fused_launches, int64_t* fused_problems);

/* Per-kernel entry points used by the parity tests (tests/test_gpu_ops.py); weights given in the reference's
 * native layouts (fp32, [Cout,Cin,kh,kw] / [N,K]) and packed on the fly.  Activations fp16 NHWC. */
int fgdm_op_conv2d(const void* x0, int C0, const void* x1, int C1, const float* w, const float* bias,
                   const float* rowvec, const void* resid, int B, int H, int W, int Cout, int ksize, int stride,
                   int upsample, int act, float scale, void* out, void* stream);
int fgdm_op_linear(const void* x, const float* w, const float* bias, const void* resid, int M, int K, int N,
                   int act, int out_kind, int rows_per_sample, int ld_out, void* out, void* stream);
/* Producer / consumer pair of a transformer-block LayerNorm (ldm/modules/attention.py:234-240: attn(norm(x)), ff(norm(x))) as
 * the engine evaluates it: h = x W1^T + b1 (+ resid) is written as fp16 [M, C] together with per-row partial sums, and
 * y = act(LayerNorm(h; gamma, beta, eps 1e-5) W2^T + b2) runs on the RAW h with the LayerNorm folded into W2 and applied to
 * the fp32 accumulator (no normalised copy of h exists).  x fp16 [M, K1], resid fp16 [M, C] or NULL, weights fp32 in the
 * reference's [out, in] layout, act2 0 or 3 (GEGLU: y is [M, N2 / 2]).  *slots_used: partial-sum slots per row. */
int fgdm_op_linear_ln_linear(const void* x, const float* w1, const float* b1, const void* resid, const float* gamma,
                             const float* beta, const float* w2, const float* b2, int M, int K1, int C, int N2, int act2,
                             void* h_out, void* y_out, int* slots_used, void* stream);
/* Tuning aids: force the implicit-GEMM tile configuration process-wide (0 = automatic; 1-3 = 2-stage kernel
 * 128x128 / 128x64 / 64x64; 4-6 = pipelined kernel 256x320 / 256x256 / 128x320), and time one conv / linear shape on
 * random data (average device milliseconds over `iters` launches). */
int fgdm_debug_force_igemm_cfg(int cfg);
int fgdm_bench_igemm(int B, int H, int W, int C0, int C1, int Cout, int ksize, int stride, int upsample, int act,
                     int use_resid, int cfg, int iters, float* avg_ms);
/* ... one attention shape (softmax(QK^T d^-1/2) V over B x heads, T queries, Tk keys) ... */
/* tools/bench_ff.py: GEGLU projection + output projection of a feed-forward block (ldm/modules/attention.py:37-64) in row
 * chunks sharing one intermediate buffer; average device milliseconds for all M rows. */
int fgdm_bench_ff(int M, int C, int chunk_rows, int iters, float* avg_ms);
int fgdm_bench_attention(int B, int heads, int T, int Tk, int d, int iters, float* avg_ms);
/* ... and one GroupNorm32(+SiLU) (kind 0, optional virtual concat C1) or LayerNorm (kind 1) shape. */
int fgdm_bench_norm(int kind, int B, int HW, int C0, int C1, int silu, int iters, float* avg_ms);
int fgdm_op_groupnorm(const void* x0, int C0, const void* x1, int C1, int B, int HW, const float* gamma,
                      const float* beta, float eps, int silu, void* out, void* stream);
int fgdm_op_layernorm(const void* x, int rows, int C, const float* gamma, const float* beta, float eps,
                      void* out, void* stream);
int fgdm_op_attention(const void* q, int ldq, const void* k, int ldk, const void* vt, int ldvt, void* o, int ldo,
                      int B, int heads, int T, int Tk, int d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
