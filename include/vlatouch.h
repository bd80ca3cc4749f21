/* vlatouch.h — C ABI of libvlatouch_hip.so, the MI355X (gfx950) engine behind the VLA-Touch
 * action-refinement path.
 *
 * The reference (jxbi1010/VLA-Touch) has no FFI: its boundary for this path is the Python class
 * surface of VLA/residual_controller/ and VLA/models/ (SURVEY.md §8b).  Those classes are mirrored in
 * vla-touch_amd/{residual_controller,models}/ and bind to the entry points below through ctypes
 * (vla-touch_amd/vlatouch/_lib.py).  Each entry cites the reference interface it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (PyTorch allocations) unless
 * marked "host"; no entry allocates device memory, synchronises the device or touches the default
 * stream — all work is enqueued on `stream` (a hipStream_t).  Return value: 0 on success, negative
 * errno-style code otherwise (-22 bad argument, -95 unsupported, -5 launch failure);
 * vt_last_error() returns a host string for the last failure on the calling thread.
 * dtype codes: 0 = fp32, 1 = bf16 (raw 16-bit), 2 = fp32 storage with split-bf16 compute (GEMM weights only), 3 = IEEE fp16.
 * "cdt" = compute/storage dtype of weights,
 * "adt" = dtype of activations between kernels.
 */
#ifndef VLATOUCH_H
#define VLATOUCH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vt_stream_t;   /* hipStream_t */
typedef struct vt_unet_s* vt_unet_t;
typedef struct vt_dino_s* vt_dino_t;
typedef struct vt_lstm_s* vt_lstm_t;
typedef struct vt_rdt_s* vt_rdt_t;

const char* vt_last_error(void);
int vt_version(void);
/* MFMA fragment-layout self test (A = I with asymmetric B, both dtypes); out_err[2] device floats. */
int vt_selftest_mfma(float* out_err, vt_stream_t stream);

/* Live timing of the LDS-DMA MFMA GEMM kernels: while enabled, every launch of the selected kernel is bracketed by HIP
 * events on its launch stream.  on = 0 off, 1 both kernels, 2 only gemm_pp256_kernel (256-square ping-pong tile),
 * 3 only gemm_glds_kernel (128-column tiles).  vt_prof_collect (after the caller synchronised the stream) returns the
 * summed duration, the algorithmic FLOPs / bytes of those launches and their count. */
int vt_prof_enable(int on);
int vt_prof_collect(double* total_ms, double* flops, double* bytes, long* launches);

/* ---------------------------------------------------------------- primitives (unit-test hooks) */
/* Generic GEMM / implicit conv1d: params = struct VtGemmParams (csrc/vt_gemm.h), host pointer.
 * Replaces torch nn.Linear / nn.Conv1d / nn.ConvTranspose1d calls of
 * bridge/networks/conditional_unet_1D.py:25,34,49,83, bridge_controller.py:42-48, HF Dinov2 linears. */
int vt_gemm(const void* params, vt_stream_t stream);
/* Frozen 16-bit weights W [N][K] (row stride ldw) -> a second copy in MFMA fragment order [N/32][K/16][64 lanes][8] (N % 32 == 0,
 * K % 16 == 0; same byte count): VtGemmParams.Wp of the weights-in-registers GEMM tile (csrc/vt_gemm_pw.hip).  Replaces nothing in
 * the reference (torch.nn.Linear keeps one layout, models/rdt/blocks.py:144-183); it is the load-time packing of this engine. */
int vt_pack_w32(const void* W, long ldw, void* out, int N, int K, vt_stream_t stream);
/* A/B tuning knobs of the GEMM dispatcher (tools/, tests): knob 1 = ring depth of the weights-in-registers tile (0 default, 4, 8);
 * knob 2 = that tile on (1) / off (0); knob 3 = the small-M tile (csrc/vt_gemm_pws.hip) on / off; knob 4 = its k-split factor (0 = choose);
 * knob 5 = timing-only ablation of the weights-in-registers tile (results are garbage: exists only in a build with -DVLATOUCH_BENCH_BUILD,
 *          tools/gemm_bench_pw.py --abl; the shipped library rejects any value but 0);
 * knob 6 = fixed-maximum softmax of the cached cross-attention on (1) / off (0: always the online form);
 * knob 7 = fused U-Net sampler path (vt_unet_fused_pack) on (1) / off (0: the launch-per-op driver);
 * knob 8 = persistent 256-square GEMM tile with the in-loop epilogue (csrc/vt_gemm_pt.hip) on (1) / off (0: gemm_pp256d_kernel);
 * knob 9 = grouped-query ViT self-attention (csrc/vt_attn.hip, attn16g_kernel: a block walks the keys once for G x 16 query rows per wave):
 *          0 off (attn16u_kernel), 1 = 64-wide heads with at most 384 query rows (DINOv2 @224; default), 3 / 6 = every 16-bit unmasked call, G pinned. */
int vt_tune(int knob, int value);

/* Flash attention, head_dim 64 (or 96: params.hd): params = struct VtAttnParams (csrc/vt_kernels.h), host pointer.
 * Replaces F.scaled_dot_product_attention (models/rdt/blocks.py:116-123) and HF Dinov2SelfAttention. */
int vt_attention(const void* params, vt_stream_t stream);
/* GroupNorm(+Mish, FiLM, residual) over fp32 split-K slabs: params = struct VtGnParams. */
int vt_groupnorm(const void* params, vt_stream_t stream);
/* Row norm: mode 0 LayerNorm, 1 RMSNorm(mean-square), 2 RMSNorm(timm<=1.0.8 unbiased-variance form). */
int vt_rownorm(const void* x, int xdt, long ldx, void* y, int ydt, long ldy, const float* w, const float* b,
               int rows, int D, float eps, int mode, vt_stream_t stream);
/* controller_dataset.py:303-346 / :349-384 (padding factor 1.4): out = (de)normalise(in) per last-dim stats. */
int vt_action_normalize(const float* in, float* out, const float* mins, const float* maxs, long n, int dim,
                        float padding_factor, int denormalize, vt_stream_t stream);
/* N(0, 1) draws on the device (counter-based Philox4x32-10 + Box-Muller) for callers that do not inject noise: replaces torch.randn
 * (rdt_runner.py:136) / torch.randn_like (bridge_model.py:372).  state = 2 x uint64 in DEVICE memory {key, next counter}, advanced behind the
 * draw by a one-thread kernel (a captured graph draws fresh noise at every replay); round_bf16 != 0 rounds the values to the bf16 grid. */
int vt_randn(float* out, long n, void* state, int round_bf16, vt_stream_t stream);
/* out [B][T][Dd] fp32 = in[:, :T, :Dd] of in [B][Tin][Din] (idt fp32 / bf16): the slice + cast between the RDT chunk and the controller's
 * `vla_actions` (frank_inference_eef.py:495-517 does it with tensor indexing). */
int vt_slice_cast(const void* in, int idt, float* out, int B, int Tin, int Din, int T, int Dd, vt_stream_t stream);
/* out [rows][cols] (odt) = in (idt), element-wise dtype conversion with row strides (fp32 / bf16 / fp16): `.to(dtype)` between pipeline
 * stages, e.g. SigLIP image tokens -> RDTRunner.predict_action's img_tokens (franka_model_eef.py:286-288). */
int vt_cast(const void* in, int idt, long ldi, void* out, int odt, long ldo, int rows, int cols, vt_stream_t stream);

/* ---------------------------------------------------------------- interpolant U-Nets + SDE sampler
 * Replaces InterpolantsConditionalUnet1D / DiffusionConditionalUnet1D.forward
 * (bridge/networks/conditional_unet_1D_si.py:4-50, conditional_unet_1D.py:194-247) and
 * StochasticInterpolants.sample / sde_vs (bridge/bridge_model.py:259-279, 334-387).
 * A handle evaluates `nets` (1 or 2) structurally identical U-Nets on the same input in grouped launches
 * (the sampler uses 2 = {v_net, s_net}); weights are packed [nets][...] per layer by the caller in the
 * order documented in csrc/vt_unet.hip (`vt_unet_weight_order`). */
typedef struct {
  int nets;            /* 1 or 2 */
  int input_dim;       /* 10 */
  int input_pad;       /* input_dim rounded up to 16 */
  int cond_dim;        /* global_cond_dim (256) */
  int dsed;            /* diffusion_step_embed_dim (256) */
  int n_groups;        /* 8 */
  int ksize;           /* 5 */
  int n_levels;        /* 3 */
  int dims[4];         /* down_dims */
  int cdt;             /* weight dtype */
  int adt;             /* activation dtype */
} vt_unet_desc;
int vt_unet_create(const vt_unet_desc* desc, const void* const* weights, int n_weights, vt_unet_t* out);
void vt_unet_destroy(vt_unet_t h);
int vt_unet_num_weights(const vt_unet_desc* desc);
size_t vt_unet_workspace_bytes(vt_unet_t h, int B, int T);
/* out[nets][B][T][input_dim] fp32 = net_i(x, t, cond); t = per-sample timesteps t_dev[B] (device) or, if
 * t_dev == NULL, the scalar t_host for every sample. */
int vt_unet_forward(vt_unet_t h, const float* x, const float* t_dev, float t_host, const float* cond,
                    float* out, int B, int T, void* workspace, vt_stream_t stream);
/* Forward velocity-score SDE (sde_vs, direction='forward', score_weight 1): x[B][T][dim] fp32 is updated in
 * place through n_steps Euler–Maruyama steps; noise = N(0,1) draws [n_steps][B][T][dim] (the reference's
 * torch.randn_like sequence) or NULL for the deterministic drift; traj (optional) receives the n_steps+1
 * states.  Schedules are the reference's string-keyed ones (bridge_model.py:59-101):
 *   gamma_type   0 '2^0.5*t(t-1)'   1 '(2t(t-1))^0.5'   2 '(1-t)^2(2t)^0.5'
 *   epsilon_type 0 '1-t'   1 't(t-1)'   2 '1-sqrt(t)'   3 '1-t^2'   4 '0'
 *   sde_type     0 'vs' (nets = {v_net, s_net}, sde_vs :334-387)   1 'bs' (nets = {b_net, s_net}, sde_bs :281-332) */
int vt_si_sample(vt_unet_t h, float* x, const float* cond, const float* noise, int n_steps, float beta_max,
                 int gamma_type, int epsilon_type, int sde_type, float* traj, int B, int T, void* workspace,
                 vt_stream_t stream);
/* The same with the two remaining arguments of sde_vs / sde_bs (bridge_model.py:281, 334): backward != 0 = direction='backward' (nets and
 * schedules evaluated at 1 - t, x <- x - (b - w eps s) dt, :356-361, :379-382; the epsilon inside b stays at t as :369 writes it),
 * score_weight = w (:376).  vt_si_sample == vt_si_sample_ex(..., 0, 1.0f, ...). */
int vt_si_sample_ex(vt_unet_t h, float* x, const float* cond, const float* noise, int n_steps, float beta_max,
                    int gamma_type, int epsilon_type, int sde_type, int backward, float score_weight, float* traj,
                    int B, int T, void* workspace, vt_stream_t stream);
/* Fused sampler path (csrc/vt_uconv.hip, vt_unet_fused.hip): in the split-bf16 mode (cdt = VT_F32X3, adt = VT_F32, dims % 64 == 0)
 * every Conv1d resolves GroupNorm + Mish + FiLM + residual of its input in its own prologue — 30 launches per SDE step instead of 67.
 * It needs a second copy of the convolution weights as pre-split bf16 hi / lo in MFMA fragment order: vt_unet_fused_bytes = size of
 * that buffer (0 = configuration not supported), vt_unet_fused_pack fills it on the device and switches vt_si_sample* (and
 * vt_unet_forward with a scalar time) of the handle to the fused path.  The buffer must outlive the handle. */
size_t vt_unet_fused_bytes(vt_unet_t h);
int vt_unet_fused_pack(vt_unet_t h, void* buf, vt_stream_t stream);
/* 1 when vt_si_sample* / vt_unet_forward of this handle take the fused path at (B, T, n_steps): packed weights present, T halves cleanly down the
 * levels (T <= 64: 16, 32, 48, 64, 24 ...; 48-tick chunks are what scripts/franka_inference_eef.py refines), every convolution of the plan finds a
 * tile and the final kernel's LDS fits the CU.  0 = the launch-per-op driver runs (same results).  vt_unet_workspace_bytes covers both. */
int vt_unet_fused_covers(vt_unet_t h, int B, int T, int n_steps);
/* Workspace bytes of the fused plan at (B, T, n_steps); 0 = the plan cannot run this shape or configuration.  Host-only (no launch, independent
 * of vt_unet_fused_pack): what vt_unet_fused_covers decides on, and part of vt_unet_workspace_bytes. */
size_t vt_unet_fused_plan_bytes(vt_unet_t h, int B, int T, int n_steps);
/* In-place per-head RMSNorm over 64-wide head slices (timm Attention q_norm/k_norm, models/rdt/blocks.py:150-156):
 * x[token*tok_stride + head*64 + 0..63], mode as vt_rownorm (1 or 2). */
int vt_headnorm(void* x, int dt, long tok_stride, int heads, long tokens, const float* w, float eps, int mode,
                vt_stream_t stream);

/* ---------------------------------------------------------------- DINOv2 CLS encoder
 * Replaces DINOv2Encoder.forward/_normalize_images (residual_controller/visual_encoder.py:56-106) and the
 * HF Dinov2Model forward it calls.  `ncams` image batches (each B images) are preprocessed with their OWN
 * max()/mean() decisions (one reference call per camera, bridge_controller.py:106-107) and then run through
 * the transformer as one batch of ncams*B. */
typedef struct {
  int hidden, layers, heads, patch, kpad;   /* kpad = 3*patch*patch rounded up to 16 (592) */
  int cdt, adt;
  float eps;
  /* ViT variants beyond DINOv2 (all 0 = DINOv2): SigLIP has no CLS token, tanh-GELU, 72-wide heads packed zero-padded to 96,
   * an FFN width that is not 4*hidden, and returns every token after the final LayerNorm */
  int no_cls;        /* 1: no CLS token (tokens = patches) */
  int act;           /* 0: erf GELU (VT_ACT_GELU_ERF); else a VT_ACT_* code; 5 (VT_ACT_SWIGLU) = dinov2-giant's gated FFN: fc1 = weights_in
                        [2*mlp_dim][hidden] -> silu(x1) * x2 -> fc2 = weights_out [hidden][mlp_dim] (HF Dinov2SwiGLUFFN) */
  int head_dim;      /* 0 / 64, or 96 (attention width heads*head_dim; weights packed accordingly) */
  int mlp_dim;       /* 0: 4*hidden; else the (64-padded) FFN width */
  int out_all;       /* 0: out = [ncams][B][hidden] CLS rows; 1: out = [ncams][B][tokens][hidden] */
  float attn_scale;  /* 0: head_dim^-0.5 */
} vt_dino_desc;
int vt_dino_create(const vt_dino_desc* desc, const void* const* weights, int n_weights, vt_dino_t* out);
void vt_dino_destroy(vt_dino_t h);
int vt_dino_num_weights(const vt_dino_desc* desc);
size_t vt_dino_workspace_bytes(vt_dino_t h, int B_total, int res);
/* Optional fragment-packed second copies of the fc1 weights (vt_pack_w32 layout; 16-bit modes, GELU FFN): the caller allocates vt_dino_packed_bytes(h) bytes
 * (0 = not applicable) and keeps them resident; vt_dino_set_packed enqueues the packing kernels.  With them the rows a 256-row tiling of the token matrix leaves
 * over, and the CLS-only last block, run fc1 on the small-M packed-weight tile.  Replaces nothing in the reference (a layout of its nn.Linear weights). */
size_t vt_dino_packed_bytes(vt_dino_t h);
int vt_dino_set_packed(vt_dino_t h, void* buf, vt_stream_t stream);
int vt_dino_set_range_flag(vt_dino_t h, unsigned* word);      /* see vt_rdt_set_range_flag */
/* imgs[ncams] device pointers; is_u8: uint8 pixels else fp32; nhwc: [B,H,W,3] else [B,3,H,W];
 * pre_scale: 1/255 when the caller passed a numpy array (visual_encoder.py:66) else 1;
 * norm_mode: 0 auto (reference behaviour), 1 force ImageNet-normalise, 2 never;
 * pos_patch: [grid*grid][hidden] fp32 position embeddings already interpolated for `res`;
 * out: [ncams][B][hidden] fp32 pooler_output; flags_out (optional) [ncams][4] = {scale, normalised, max, mean}. */
int vt_dino_forward(vt_dino_t h, const void* const* imgs, int ncams, int is_u8, int nhwc, float pre_scale,
                    int norm_mode, int B, int res, const float* pos_patch, float* out, float* flags_out,
                    void* workspace, vt_stream_t stream);

/* ---------------------------------------------------------------- small MLP chain (state encoder / force encoder / heads)
 * y = L_n(...act(L_1(x))): replaces the nn.Sequential MLPs of bridge_controller.py:42-48 and
 * lstm_step_controller.py:44-60.  W_i [out_i][in_pad_i] cdt (in padded to 16), b_i fp32.  x is [B][ldx] of adt
 * with zero padding up to in_pad_1.  tmp: 2*B*max(out_i) adt elements. */
int vt_mlp(const void* x, long ldx, int B, int n_layers, const int* dims /* n_layers+1, padded ins */,
           const void* const* W, const float* const* b, int act, void* y, int ydt, long ldy,
           int cdt, int adt, void* tmp, vt_stream_t stream);
/* Gather [cls_cam1 | cls_cam2 | state | forces] rows into a zero-padded adt matrix (bridge_controller.py:129-132). */
int vt_concat_obs(const float* cls1, const float* cls2, int dv, const float* state, int sdim, const float* forces,
                  int fdim, void* out, int odt, long ldo, int B, vt_stream_t stream);

/* ---------------------------------------------------------------- LSTM residual head
 * Replaces TactileLSTMController.predict / predict_sequence / forward (lstm_step_controller.py:232-286, 288-319, 170-213):
 * force MLP -> L-layer LSTM cell with carried (h, c) -> [h_top | obs_cond] head -> vla_n + delta, as ONE persistent kernel for T
 * consecutive ticks (csrc/vt_lstm.hip).  hidden must be 256; weights pre-packed in MFMA fragment order by the host side
 * (order and layout: csrc/vt_lstm.hip above vt_lstm_create).  force_pad / in_pad are kept for ABI stability and ignored. */
typedef struct { int state_dim, hidden, layers, force_dim, force_pad, in_pad, cdt; } vt_lstm_desc;
int vt_lstm_create(const vt_lstm_desc* desc, const void* const* weights, int n_weights, vt_lstm_t* out);
void vt_lstm_destroy(vt_lstm_t h);
int vt_lstm_num_weights(const vt_lstm_desc* desc);
size_t vt_lstm_workspace_bytes(vt_lstm_t h, int B);
/* One tick: out_n[B][state_dim] = vla_n + delta (normalised); h, c: [layers][B][hidden] fp32, updated in place. */
int vt_lstm_step(vt_lstm_t hd, const float* obs_cond, const float* vla_n, const float* force, float* h, float* c,
                 float* out_n, int B, void* workspace, vt_stream_t stream);
/* T ticks in one launch: vla_n [B][T][state_dim], force [B][T][force_dim] -> out_n [B][T][state_dim]; (h, c) carried in LDS /
 * registers across the ticks, read at entry and written back at exit. */
int vt_lstm_sequence(vt_lstm_t hd, const float* obs_cond, const float* vla_n, const float* force, float* h, float* c,
                     float* out_n, int B, int T, vt_stream_t stream);

/* ---------------------------------------------------------------- RDT diffusion transformer + DPM-Solver++ sampler
 * Replaces RDT.forward (models/rdt/model.py:126-165; blocks models/rdt/blocks.py:72-202) and
 * RDTRunner.predict_action / conditional_sample / adapt_conditions (models/rdt_runner.py:108-165, 225-250).
 * head_dim must be 64; rms_mode as vt_rownorm (1 = mean-square, 2 = timm<=1.0.8 variance form).
 * Weight order: csrc/vt_rdt.hip header. */
typedef struct {
  int hidden, depth, heads, horizon, out_dim;
  int state_dim;        /* state_token_dim (128); the state adaptor takes 2*state_dim (state | mask) */
  int lang_dim, img_dim;
  int max_lang_len, img_len;
  int n_lang, n_img, n_state;   /* adaptor depths: 1 = 'linear', N = 'mlpNx_gelu' */
  int cdt, adt;                 /* both fp32 or both bf16 */
  int rms_mode;
} vt_rdt_desc;
int vt_rdt_create(const vt_rdt_desc* desc, const void* const* weights, int n_weights, vt_rdt_t* out);
void vt_rdt_destroy(vt_rdt_t h);
int vt_rdt_num_weights(const vt_rdt_desc* desc);
size_t vt_rdt_workspace_bytes(vt_rdt_t h, int B, int lang_len);
/* Optional fragment-packed second copies of the Linears of the denoise loop (vt_pack_w32 layout; bf16, hidden % 512 == 0): the caller
 * allocates vt_rdt_packed_bytes(h) bytes (0 = not applicable), vt_rdt_set_packed enqueues the packing kernels and keeps the pointers. */
/* bounds[depth] (host): per block an upper bound of |q . k| * scale of its cross-attention (8 max|q_norm.weight| max|k_norm.weight| for the
 * mean-square RMSNorm of blocks.py:86-87; 0 = none) -> the cached cross-attention uses a fixed-maximum softmax where the bound is <= 40. */
int vt_rdt_set_score_bounds(vt_rdt_t h, const float* bounds, int n);
/* 16-bit mode only: 1 (default) = DPM-Solver++ state, x0 predictions and the final projection in fp32 between network evaluations; 0 = the
 * reference's bf16 rounding points (models/rdt_runner.py:137-139,160: `noisy_action.to(dtype)` after every scheduler step). */
int vt_rdt_set_state_precision(vt_rdt_t h, int fp32_state);
/* 16-bit modes: the reference's dtype when it differs from the engine's compute type (desc.cdt = desc.adt = VT_F16 evaluating a bf16 model with IEEE
 * fp16 activations — same width and MFMA rate, 3 more mantissa bits, the bf16 weights convert exactly): the start noise (models/rdt_runner.py:137-139)
 * and, with vt_rdt_set_state_precision(h, 0), the solver state are rounded to THIS grid.  VT_BF16 (default for bf16 engines) or VT_F16. */
int vt_rdt_set_io_dtype(vt_rdt_t h, int io_dtype);
size_t vt_rdt_packed_bytes(vt_rdt_t h);
int vt_rdt_set_packed(vt_rdt_t h, void* buf, vt_stream_t stream);
/* Range guard (round 6).  The reference executes RDT in bf16 (models/rdt_runner.py:47-60,160; models/rdt/model.py:124: 8 exponent bits); an engine created
 * with desc.cdt = desc.adt = VT_F16 has 5.  `word` = ONE zero-initialised uint32 in device memory, owned by the caller, resident as long as the handle is used
 * (NULL = detach).  The kernels of vt_rdt_sample / vt_rdt_forward OR bits into it — no extra launch, never cleared by the library, so it is sticky across calls
 * and hipGraph replays; the caller reads it when it likes (after a stream synchronise) and re-zeroes it:
 *   VT_RANGE_XN_SAT     the un-normalised 16-bit operand x * gain a residual Linear hands to the next Linear (csrc/vt_gemm_pw.hip) left the fp16 range and was clamped
 *   VT_RANGE_NONFINITE  an x0 prediction / solver state (vt_rdt_sample) or an output element (vt_rdt_forward) is inf or NaN: every other 16-bit store of the path
 *                       converts with v_cvt_f16_f32 (overflow -> inf), and inf / NaN propagate through every consumer (GEMMs, norms, softmax) into the fp32
 *                       residual stream, so an overflow anywhere upstream — inputs and converted weights included — ends here
 *   VT_RANGE_ATTN_EMPTY a cross-attention row whose probabilities summed to 0 or inf (all keys masked, or a degenerate score row); the row is written as zeros
 *                       (torch's SDPA returns NaN for a fully masked row, models/rdt/blocks.py:116-123)
 * vt_dino_set_range_flag: the same word for a DINOv2 / SigLIP handle — VT_RANGE_GATE_SAT: the gated SwiGLU product of dinov2-giant left the fp16 range and was clamped. */
#ifndef VT_RANGE_XN_SAT
#define VT_RANGE_XN_SAT 1u
#define VT_RANGE_NONFINITE 2u
#define VT_RANGE_GATE_SAT 4u
#define VT_RANGE_ATTN_EMPTY 8u
#endif
int vt_rdt_set_range_flag(vt_rdt_t h, unsigned* word);
/* RDT.forward: x_tokens [B][horizon+1][hidden] adt (adapted state + action tokens), freq [B] fp32, t = t_dev[B] or the
 * scalar t_host when t_is_scalar, lang_c [B][L][hidden] / img_c [B][img_len][hidden] adt (adapted, before position
 * embeddings), lang_mask [B][L] bytes (1 = valid) or NULL -> out [B][horizon][out_dim] adt. */
int vt_rdt_forward(vt_rdt_t h, const void* x_tokens, const float* freq, const float* t_dev, float t_host, int t_is_scalar,
                   const void* lang_c, const void* img_c, const uint8_t* lang_mask, void* out, int B, int L,
                   void* workspace, vt_stream_t stream);
/* RDTRunner.predict_action: lang_tokens [B][L][lang_dim], img_tokens [B][img_len][img_dim], state_tokens [B][1][state_dim],
 * action_mask [B][1][state_dim] (all adt), ctrl_freqs [B] fp32, x_init [B][horizon][state_dim] fp32 (the N(0,1) start the
 * reference draws with torch.randn), host arrays timesteps[n_steps] and coef[n_steps][5] = {a, b0, b1, alpha_s, sigma_s} of
 * the multistep update x <- a x + b0 x0_k + b1 x0_{k-1}; sample_pred: 1 = 'sample' prediction, 0 = 'epsilon'.
 * adapted != 0 = conditional_sample (rdt_runner.py:122): lang/img/state tokens are already adapted to [.., hidden].
 * out [B][horizon][state_dim] fp32 (values on the adt grid, masked). */
int vt_rdt_sample(vt_rdt_t h, const void* lang_tokens, const uint8_t* lang_mask, const void* img_tokens,
                  const void* state_tokens, const void* action_mask, const float* ctrl_freqs, const float* x_init,
                  int n_steps, const float* timesteps, const float* coef, int sample_pred, int adapted, float* out,
                  int B, int L, void* workspace, vt_stream_t stream);

/* ---------------------------------------------------------------- GelSight marker tracker -> m_t
 * Replaces EnhancedMarkerTracker.init_standard / detect_markers / match_and_compute_displacement / estimate_force
 * (residual_controller/tactile/marker/marker_tracker.py:81-114, 154-183, 308-341, 343-373) for a BATCH of frames.
 * frames [N][H][W][channels] bytes (channels 3 = BGR as cv2 delivers it, 1 = gray).  Per frame: 5x5 blur -> adaptive
 * Gaussian threshold (11, C = 2, inverted) -> 3x3 open -> outer contours of the 8-connected components -> polygon area in
 * (min_area, max_area) -> truncated polygon centroid, in cv2.findContours order (last found first).
 * markers [N][max_markers][2] int32 (x, y), counts [N] (may exceed max_markers or max_cand: overflow, the caller checks),
 * binary_out [N][H][W] bytes (0/1; the reference's `processed_frame` / 255) or NULL.
 * mode 0: gelsight_version 'standard' (init_standard).  mode 1: `frames` is already a processed binary image [N][H][W] (non-zero =
 * marker), only the contour stage runs (detect_markers :154).  mode 2: 'HSR' (init_HSR :116-152: invert, equalizeHist, 5x5 blur,
 * threshold > 50, 3x3 open).  mode | 0x100: BGR2GRAY with the 14-bit coefficient table of OpenCV <= 3.4.1 ((1868 B + 9617 G + 4899 R +
 * 8192) >> 14) instead of the 15-bit set of OpenCV >= 3.4.2 / 4.x ((3735 B + 19235 G + 9798 R + 16384) >> 15, the default: what the
 * reference's unpinned `opencv-python` resolves to). */
size_t vt_marker_workspace_bytes(int N, int H, int W, int max_cand);
int vt_marker_detect(const uint8_t* frames, int channels, int mode, int N, int H, int W, double min_area, double max_area,
                     int max_cand, int* markers, int* counts, int max_markers, uint8_t* binary_out, void* workspace,
                     vt_stream_t stream);
/* baseline [n_base][2] int32 -> disp [N][max_markers][2] int32 (marker - nearest baseline marker, ties to the lower index),
 * force [N][3] fp64 = |mean displacement|, unit direction x, y (zeros when no marker). */
int vt_marker_displacement(const int* markers, const int* counts, int N, int max_markers, const int* baseline, int n_base,
                           int* disp, double* force, vt_stream_t stream);

/* ---------------------------------------------------------------- controller training step primitives
 * Replace torch autograd / optim.AdamW / torch_ema around the interpolant losses (bridge/bridge_model.py:183-246 velocity_loss,
 * score_loss, b_loss, get_loss; bridge_train.py:49-58, 312-334).  fp32, channel-last [B][T][C]; the matrix products of the backward pass
 * are vt_gemm calls on the buffers these produce (vlatouch/train.py composes them; csrc/vt_train.hip). */
/* out[(tap*Cin + ci)][b*Tout + t] = x[b][t*stride + off0 + tap][ci] (0 outside): transposed im2col, the operand of a weight gradient */
int vt_im2col_t(const float* x, float* out, int B, int Tin, int Tout, int Cin, int taps, int stride, int off0, vt_stream_t stream);
int vt_transpose(const float* in, float* out, int M, int N, vt_stream_t stream);                 /* [M][N] -> [N][M] */
int vt_zero_stuff(const float* x, float* out, int B, int T, int C, vt_stream_t stream);           /* out[b][2u] = x[b][u], odd rows 0 */
int vt_wflip(const float* W, float* WT, int Cout, int taps, int Cin, vt_stream_t stream);         /* [Cout][taps][Cin] -> [Cin][taps reversed][Cout] */
int vt_colsum(const float* x, long ld, float* out, int M, int N, int accumulate, vt_stream_t stream);
int vt_add_(float* a, const float* b, long n, vt_stream_t stream);
int vt_copy_cols(const float* src, long lds, int off, float* dst, long ldd, int doff, int rows, int cols, int accumulate, vt_stream_t stream);
int vt_mish(const float* x, const float* dy, float* out, long n, vt_stream_t stream);             /* dy NULL: mish(x); else dy * mish'(x) */
/* backward of out = film_scale * mish(GroupNorm(c)) + film_bias: c, dout, dc [B*T][C]; film, dfilm [B][2C] (scale | bias) or NULL;
 * dgamma_part, dbeta_part [B][C] per-sample partials (sum over B with vt_colsum). */
int vt_gn_mish_bwd(const float* c, const float* gamma, const float* beta, const float* film, const float* dout, float* dc,
                   float* dgamma_part, float* dbeta_part, float* dfilm, int B, int T, int C, int ngroups, float eps, vt_stream_t stream);
int vt_gelu(const float* x, const float* dy, float* out, long n, vt_stream_t stream);             /* erf GELU / its backward */
/* q_sample + the three loss targets of the LINEAR interpolant (bridge_model.py:103-107, 183-217, 248-258): xt, target_v = x1 - x0,
 * target_s = -z, target_b = (x1 - x0) + gamma'(t) z, t_clipped[B]; z already scaled by beta_max; gamma_type as vt_si_sample. */
int vt_si_qsample(const float* x0, const float* x1, const float* z, const float* t, float* xt, float* target_v, float* target_s,
                  float* target_b, float* t_clipped, int B, long per_sample, int gamma_type, float t_min, vt_stream_t stream);
/* the same for every `interpolant_type` of the reference (bridge_model.py:103-181: xt = w0(t) x0 + w1(t) x1 + gamma z, target_v = d(w0 x0 + w1 x1) / dt,
 * target_b = target_v + gamma'(t) z): 0 linear, 1 power3, 2 power4, 3 reverse_power3, 4 reverse_power4, 5 gaussian_encode_decode, 6 reverse_linear */
int vt_si_qsample_ex(const float* x0, const float* x1, const float* z, const float* t, float* xt, float* target_v, float* target_s,
                     float* target_b, float* t_clipped, int B, long per_sample, int gamma_type, float t_min, int interpolant_type, vt_stream_t stream);
/* loss[0] = mean_b(0.5 |out_b|^2 - <target_b, out_b>), dout = (out - target) / B   (the three interpolant losses share this form) */
int vt_si_loss(const float* out, const float* target, float* dout, float* loss, int B, long per_sample, vt_stream_t stream);
int vt_slab_sum(const float* slabs, int S, long n, const float* bias, int N, float* out, vt_stream_t stream);  /* split-K partials [S][n] -> out[n] (+ bias[i % N]) */
int vt_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
             int step, vt_stream_t stream);
int vt_ema_update(float* shadow, const float* p, long n, float decay, vt_stream_t stream);
/* the same two updates with the step-dependent scalars in DEVICE memory, hyper = [lr, 1 - beta1^t, sqrt(1 - beta2^t), 1 - ema_decay_t]:
 * a captured graph of the training step stays valid while t advances (the host rewrites 16 bytes before each replay) */
int vt_train_hyper(float lr, float beta1, float beta2, int step, float ema_decay, float* out4_host);   /* host-side: fills hyper for step t */
int vt_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2, float eps,
                 float weight_decay, vt_stream_t stream);
/* AdamW (+ EMA) over a device table of tensors in ONE launch.  table = ntensors records of 7 x 8 bytes:
 * {float* p, const float* g, float* m, float* v, float* shadow_or_null, int64 n, int64 first_chunk}, first_chunk = running sum of
 * ceil(n / 4096) over the preceding records; total_chunks = that sum over all records.  hyper as vt_adamw_dev. */
int vt_adamw_ema_multi(const void* table, int ntensors, long total_chunks, const float* hyper, float beta1, float beta2, float eps,
                       float weight_decay, vt_stream_t stream);
int vt_ema_update_dev(float* shadow, const float* p, long n, const float* hyper, vt_stream_t stream);
int vt_posemb(const float* t, float* out, int B, int dim, vt_stream_t stream);                    /* SinusoidalPosEmb: [sin | cos] */

/* ---- LSTM residual head training (lstm_step_controller.py:176-211 forward, :321-337 get_loss; lstm_train.py:26-33, 129-133).
 * Batch-major sequences [B][T][C]; `t` is the tick a call works on.  Gate order i, f, g, o (torch.nn.LSTM).
 * vt_lstm_cell_fwd: a = gx[b][t] (+ gh[b], required for t > 0) -> act[b][t][4H] (activated gates), cseq[b][t], hseq[b][t],
 *                   hprev[b][t+1] (= h_t; hprev[b][0] = 0), hcur[b] (contiguous copy of h_t).
 * vt_lstm_cell_bwd: dh = dhseq[b][t] (+ dh_rec[b], required for t < T-1); writes pre-activation gradients to dgates[b][t][4H] and
 *                   dgcur[b][4H]; dc_next[b] carries dL/dc between ticks (ignored on input at t = T-1).
 * vt_ln_bwd:        LayerNorm backward: dx, and dyxh = dy * x_hat whose column sums are d gamma (d beta = column sums of dy).
 * vt_bcast_mid / vt_sum_mid: dst[b][t][doff + c] = src[b][c] and its gradient out[b][c] = sum_t src[b][t][off + c].
 * vt_mse_residual:  pred = base + delta (base may be null), loss = mean((pred - target)^2), ddelta = 2 (pred - target) / n. */
int vt_lstm_cell_fwd(const float* gx, const float* gh, float* act, float* cseq, float* hseq, float* hprev, float* hcur, int B, int T, int H,
                     int t, vt_stream_t stream);
int vt_lstm_cell_bwd(const float* dhseq, const float* dh_rec, const float* act, const float* cseq, float* dc_next, float* dgates, float* dgcur,
                     int B, int T, int H, int t, vt_stream_t stream);
int vt_ln_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dyxh, int rows, int C, float eps, vt_stream_t stream);
int vt_bcast_mid(const float* src, float* dst, long ldd, int doff, int B, int T, int C, vt_stream_t stream);
int vt_sum_mid(const float* src, long lds, int off, float* out, int B, int T, int C, vt_stream_t stream);
int vt_mul_(float* a, const float* b, long n, vt_stream_t stream);
int vt_mse_residual(const float* base, const float* delta, const float* target, float* pred, float* ddelta, float* loss, long n, vt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
