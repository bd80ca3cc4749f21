// vt_models.hip — host drivers: DINOv2 CLS encoder, small MLP chains, observation concat.
//
// DINOv2 weight order (vt_dino_create):
//   0 patch_w [D][kpad] cdt   1 patch_b [D]   2 cls_pos0 [D] (= cls_token + position_embeddings[0])
//   per layer (14 entries): ln1_w ln1_b  qkv_w [3D][D] cdt  qkv_b [3D]  proj_w [D][D] cdt  proj_b  ls1
//                           ln2_w ln2_b  fc1_w [4D][D] cdt  fc1_b  fc2_w [D][4D] cdt  fc2_b  ls2
//                           (act = VT_ACT_SWIGLU, dinov2-giant: fc1_w = weights_in [2F][D], fc2_w = weights_out [D][F], F = mlp_dim)
//   then  lnf_w  lnf_b
// The residual stream (tokens) is kept in fp32; normalised activations / QKV / MLP hidden are `adt`.
//
// (The LSTM residual head is one persistent kernel: vt_lstm.hip.)
#include <math.h>
#include <string.h>
#include <new>
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_host.h"
#include "vt_prof.h"
#include "../../include/vlatouch.h"

#define CK(x) do { int _r = (x); if (_r) return _r; } while (0)
static int es(int dt) { return (dt == VT_BF16 || dt == VT_F16) ? 2 : 4; }

static VtGemmParams lin(const void* A, int adt, long lda, const void* W, int cdt, long ldw, const float* b, void* C, int odt, long ldc,
                        int M, int N, int K, int act) {
  VtGemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
  p.bias = b; p.act = act; p.groups = 1; p.splitk = 1; p.a_dtype = adt; p.w_dtype = cdt; p.c_dtype = odt;
  return p;
}

// ======================================================================================= DINOv2
struct DinoLayer {
  const float *ln1_w, *ln1_b, *qkv_b, *proj_b, *ls1, *ln2_w, *ln2_b, *fc1_b, *fc2_b, *ls2;
  const void *qkv_w, *proj_w, *fc1_w, *fc2_w;
  const void* fc1_wp;      // optional fragment-packed copy of fc1 (vt_dino_set_packed): the <= 64-row remainder of fc1's exact row split runs on the small-M packed-weight tile
};
struct vt_dino_s {
  vt_dino_desc d;
  const void* patch_w; const float *patch_b, *cls_pos0, *lnf_w, *lnf_b;
  DinoLayer L[48];
  unsigned* range_flag = nullptr;      // range guard word (vt_dino_set_range_flag)
};

int vt_dino_num_weights(const vt_dino_desc* d) { return 3 + 14 * d->layers + 2; }
int vt_dino_set_range_flag(vt_dino_t h, unsigned* word) {
  if (!h) return vt_fail(VT_ERR_ARG, "vt_dino_set_range_flag: null handle");
  h->range_flag = word;
  return VT_OK;
}

int vt_dino_create(const vt_dino_desc* desc, const void* const* w, int n, vt_dino_t* out) {
  if (!desc || !w || !out) return vt_fail(VT_ERR_ARG, "vt_dino_create: null argument");
  const vt_dino_desc& d = *desc;
  const int hd = d.head_dim ? d.head_dim : 64;
  if (d.layers < 1 || d.layers > 48 || d.hidden % 64 || (hd != 64 && hd != 96 && !(hd == 80 && d.adt != VT_F32)) || (hd == 64 && d.hidden / d.heads != 64) || d.patch != 14 ||
      d.kpad % 16 || d.kpad < 588 || (d.mlp_dim % 16))
    return vt_fail(VT_ERR_ARG, "vt_dino_create: unsupported config (head_dim 64 / 96, or 80 in a 16-bit mode; patch 14)");
  if (n != vt_dino_num_weights(desc)) return vt_fail(VT_ERR_ARG, "vt_dino_create: expected %d weights, got %d", vt_dino_num_weights(desc), n);
  for (int k = 0; k < n; ++k) if (!w[k]) return vt_fail(VT_ERR_ARG, "vt_dino_create: weight %d is null", k);
  vt_dino_s* h = new (std::nothrow) vt_dino_s();
  if (!h) return vt_fail(-12, "out of host memory");
  h->d = d;
  int i = 0;
  h->patch_w = w[i++]; h->patch_b = (const float*)w[i++]; h->cls_pos0 = (const float*)w[i++];
  for (int l = 0; l < d.layers; ++l) {
    DinoLayer& L = h->L[l];
    L.ln1_w = (const float*)w[i++]; L.ln1_b = (const float*)w[i++]; L.qkv_w = w[i++]; L.qkv_b = (const float*)w[i++];
    L.proj_w = w[i++]; L.proj_b = (const float*)w[i++]; L.ls1 = (const float*)w[i++];
    L.ln2_w = (const float*)w[i++]; L.ln2_b = (const float*)w[i++]; L.fc1_w = w[i++]; L.fc1_b = (const float*)w[i++];
    L.fc2_w = w[i++]; L.fc2_b = (const float*)w[i++]; L.ls2 = (const float*)w[i++];
  }
  h->lnf_w = (const float*)w[i++]; h->lnf_b = (const float*)w[i++];
  *out = h;
  return VT_OK;
}
void vt_dino_destroy(vt_dino_t h) { delete h; }

// Optional fragment-packed second copies of the fc1 weights (vt_pack_w32 layout; 16-bit, GELU FFN, hidden % 256 == 0, mlp % 64 == 0): with them the few rows a
// 256-row tiling of B x N tokens leaves over (DINOv2-base, 64 images: 64 x 257 = 64 x 256 + 64) and the CLS-only last block take vt_gemm_pws.hip (5 us) instead of
// the register-staged generic GEMM (18 us, 12 launches per forward).  The caller owns `buf` (vt_dino_packed_bytes(h) bytes, resident while the handle is used).
static bool dino_pk_ok(const vt_dino_s* h) {
  const int D = h->d.hidden, Dm = h->d.mlp_dim ? h->d.mlp_dim : 4 * D;
  return h->d.cdt != VT_F32 && h->d.adt == h->d.cdt && h->d.act != VT_ACT_SWIGLU && D % 256 == 0 && Dm % 64 == 0;
}
size_t vt_dino_packed_bytes(vt_dino_t h) {
  if (!h || !dino_pk_ok(h)) return 0;
  const size_t D = h->d.hidden, Dm = h->d.mlp_dim ? h->d.mlp_dim : 4 * D;
  return (size_t)h->d.layers * Dm * D * 2;
}
int vt_dino_set_packed(vt_dino_t h, void* buf, vt_stream_t stream) {
  if (!h) return vt_fail(VT_ERR_ARG, "vt_dino_set_packed: null handle");
  if (!vt_dino_packed_bytes(h)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_dino_set_packed: this configuration has no packed weights");
  if (!buf) return vt_fail(VT_ERR_ARG, "vt_dino_set_packed: null buffer");
  const int D = h->d.hidden, Dm = h->d.mlp_dim ? h->d.mlp_dim : 4 * D;
  char* o = (char*)buf;
  for (int l = 0; l < h->d.layers; ++l) {
    const int r = vt_pack_w32(h->L[l].fc1_w, D, o, Dm, D, stream);
    if (r) return r;
    h->L[l].fc1_wp = o;
    o += (size_t)Dm * D * 2;
  }
  return VT_OK;
}

namespace {
constexpr int DINO_SPLITK = 4, DINO_SPLIT_ROWS = 1100;
struct DWs { size_t part, flags, apatch, tok, xn, qkv, att, h1, slab, slab_bytes, total; };
DWs dcarve(const vt_dino_s* h, int Bt, int res) {
  const vt_dino_desc& d = h->d;
  const int a = es(d.adt), g = res / d.patch, N = g * g + (d.no_cls ? 0 : 1), D = d.hidden;
  const int Da = d.heads * (d.head_dim ? d.head_dim : 64), Dm = d.mlp_dim ? d.mlp_dim : 4 * D;
  DWs w; size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += (b + 255) / 256 * 256; return r; };
  w.part = take(512 * 4 * 8); w.flags = take(8 * 4 * 8);
  w.apatch = take((size_t)Bt * g * g * d.kpad * a);
  w.tok = take((size_t)Bt * N * D * 4);
  w.xn = take((size_t)Bt * N * D * a);
  w.qkv = take((size_t)Bt * N * 3 * Da * a);
  w.att = take((size_t)Bt * N * Da * a);
  w.h1 = take((size_t)Bt * N * Dm * a * (d.act == VT_ACT_SWIGLU ? 2 : 1));      // SwiGLU: fc1 produces [x1 | x2]
  // split-K scratch of fc2 at a few images (batch 1 of the robot loop: 2 x 257 rows): DINO_SPLITK fp32 slabs of [rows][D]
  w.slab_bytes = (size_t)Bt * N <= DINO_SPLIT_ROWS ? (size_t)DINO_SPLITK * Bt * N * D * 4 : 0;
  w.slab = take(w.slab_bytes);
  w.total = o;
  return w;
}
}  // namespace

size_t vt_dino_workspace_bytes(vt_dino_t h, int B_total, int res) { return h ? dcarve(h, B_total, res).total : 0; }

int vt_dino_forward(vt_dino_t h, const void* const* imgs, int ncams, int is_u8, int nhwc, float pre_scale, int norm_mode, int B, int res,
                    const float* pos_patch, float* out, float* flags_out, void* workspace, vt_stream_t stream) {
  if (!h || !imgs || !pos_patch || !out || !workspace) return vt_fail(VT_ERR_ARG, "vt_dino_forward: null argument");
  if (ncams < 1 || ncams > 8 || B < 1 || res < 14) return vt_fail(VT_ERR_ARG, "vt_dino_forward: bad sizes");
  const vt_dino_desc& d = h->d;
  hipStream_t s = (hipStream_t)stream;
  const int g = res / d.patch, np = g * g, cls = d.no_cls ? 0 : 1, N = np + cls, D = d.hidden, Bt = ncams * B, a = es(d.adt);
  const int hd = d.head_dim ? d.head_dim : 64, Da = d.heads * hd, Dm = d.mlp_dim ? d.mlp_dim : 4 * D;
  const int act = d.act ? d.act : VT_ACT_GELU_ERF;
  const float scale = d.attn_scale > 0.f ? d.attn_scale : 1.0f / sqrtf((float)hd);
  char* ws = (char*)workspace;
  const DWs w = dcarve(h, Bt, res);
  float* tok = (float*)(ws + w.tok);
  // 1. per-camera statistics -> branch flags; patchify with the camera's own decision
  for (int c = 0; c < ncams; ++c) {
    float* part = (float*)(ws + w.part) + c * 512;
    float* flags = (float*)(ws + w.flags) + c * 8;
    const long n = (long)B * 3 * res * res;
    CK(vt_wrap(vt_k_imgstats(imgs[c], is_u8, n, pre_scale, norm_mode, part, flags, s, flags_out ? flags_out + c * 4 : nullptr), "dino imgstats"));
    CK(vt_wrap(vt_k_patchify(imgs[c], is_u8, nhwc, B, res, g, d.kpad, flags, ws + w.apatch + (size_t)c * B * np * d.kpad * a, d.adt, s), "dino patchify"));
  }
  // 2. patch-embed GEMM, one group per image, + position embedding (shared residual) -> fp32 tokens rows 1..np
  {
    VtGemmParams p = lin(ws + w.apatch, d.adt, d.kpad, h->patch_w, d.cdt, d.kpad, h->patch_b, tok + (size_t)cls * D, VT_F32, D, np, D, d.kpad, VT_ACT_NONE);
    p.groups = Bt; p.a_gs = (long)np * d.kpad; p.w_gs = 0; p.c_gs = (long)N * D; p.bias_gs = 0;
    p.residual = pos_patch; p.ldr = D; p.r_gs = 0;
    CK(vt_wrap(vt_gemm_launch(p, s), "dino patch embed"));
    if (cls) CK(vt_k_bcast_row(h->cls_pos0, tok, (long)N * D, Bt, D, s));
  }
  const int M = Bt * N;
  // The reference consumes only pooler_output = the CLS row of the last layer (visual_encoder.py:88-93): in the LAST block every token still feeds
  // K and V, but only the CLS row needs a query, the output projection, the MLP and the residual updates (SURVEY 2.1 "last layer prunes to the CLS
  // query").  VLATOUCH_DINO_CLS_LAST=0 runs the last block on all tokens (A/B; same CLS row up to the summation order of the small-M GEMM tiles).
  // the FFN of a block on `rows` rows of xn (row stride D) -> tok rows (row stride tstride_out): fc1 (+ activation, or the SwiGLU gate) -> fc2 + LayerScale + residual
  const bool swiglu = act == VT_ACT_SWIGLU;
  auto ffn = [&](const DinoLayer& L, int rows, long tok_stride) -> int {
    const int N1 = swiglu ? 2 * Dm : Dm;
    { VtGemmParams p = lin(ws + w.xn, d.adt, D, L.fc1_w, d.cdt, D, L.fc1_b, ws + w.h1, d.adt, N1, rows, N1, D, swiglu ? VT_ACT_NONE : act);
      p.Wp = swiglu ? nullptr : L.fc1_wp;
      if (p.Wp && vt_gemm_fast_eligible(p) && vt_gemm_pw_eligible(p)) p.Wp = nullptr;   // the 160 x 128 weights-in-registers tile is the RDT denoise loop's; ViT GEMMs stay on the persistent tile
      CK(vt_wrap(vt_gemm_launch(p, s), "dino fc1")); }
    if (swiglu) CK(vt_wrap(vt_k_swiglu(ws + w.h1, d.adt, N1, rows, Dm, s, h->range_flag), "dino swiglu gate"));
    { VtGemmParams p = lin(ws + w.h1, d.adt, N1, L.fc2_w, d.cdt, Dm, L.fc2_b, tok, VT_F32, tok_stride, rows, D, Dm, VT_ACT_NONE);
      p.colscale = L.ls2; p.residual = tok; p.ldr = tok_stride;
      // a few images (2 x 257 rows at batch 1): 108 tiles of 64 x 64 would each walk K = 3072 alone (50 us); DINO_SPLITK slices per tile into fp32
      // slabs + the slab reduction (bias, LayerScale, residual) take 17
      if (w.slab_bytes && rows > 64 && (size_t)DINO_SPLITK * rows * D * 4 <= w.slab_bytes && Dm >= 2048 && (Dm / 64) % DINO_SPLITK == 0 && D % 4 == 0 &&
          !vt_gemm_fast_eligible(p)) {
        VtGemmParams q = p;
        q.C = ws + w.slab; q.c_dtype = VT_F32; q.ldc = D; q.splitk = DINO_SPLITK; q.c_slab = (long)rows * D;
        q.bias = nullptr; q.colscale = nullptr; q.residual = nullptr;
        CK(vt_wrap(vt_gemm_launch(q, s), "dino fc2 (split)"));
        CK(vt_wrap(vt_k_slab_reduce((const float*)(ws + w.slab), DINO_SPLITK, q.c_slab, rows, D, p.bias, VT_ACT_NONE, p.colscale, p.residual, p.ldr, p.C, VT_F32,
                                    tok_stride, nullptr, nullptr, 0, 0, 0.f, 0, s), "dino fc2 (slab reduction)"));
      } else {
        CK(vt_wrap(vt_gemm_launch(p, s), "dino fc2"));
      } }
    return VT_OK;
  };
  static const bool cls_last_on = [] { const char* e = getenv("VLATOUCH_DINO_CLS_LAST"); return !e || atoi(e) != 0; }();
  for (int l = 0; l < d.layers; ++l) {
    const DinoLayer& L = h->L[l];
    const bool cls_only = cls_last_on && cls && !d.out_all && l == d.layers - 1;
    CK(vt_k_rownorm(tok, VT_F32, D, ws + w.xn, d.adt, D, L.ln1_w, L.ln1_b, M, D, d.eps, VT_NORM_LAYER, s));
    if (cls_only) {
      const long tstride = (long)N * D;                 // CLS row of image b = row b * N of the token matrix
      // K | V of every token (rows Da .. 3 Da of the fused qkv weight) + Q of the CLS rows only
      { VtGemmParams p = lin(ws + w.xn, d.adt, D, (const char*)L.qkv_w + (size_t)Da * D * es(d.cdt), d.cdt, D, L.qkv_b + Da, ws + w.qkv + (size_t)Da * a, d.adt, 3 * Da,
                             M, 2 * Da, D, VT_ACT_NONE);
        CK(vt_wrap(vt_gemm_launch(p, s), "dino kv (last block)")); }
      { VtGemmParams p = lin(ws + w.xn, d.adt, tstride, L.qkv_w, d.cdt, D, L.qkv_b, ws + w.qkv, d.adt, (long)N * 3 * Da, Bt, Da, D, VT_ACT_NONE);
        CK(vt_wrap(vt_gemm_launch(p, s), "dino q (CLS rows)")); }
      { VtAttnParams p;
        memset(&p, 0, sizeof(p));
        p.Q = ws + w.qkv; p.K = ws + w.qkv + (size_t)Da * a; p.V = ws + w.qkv + (size_t)2 * Da * a; p.O = ws + w.att;
        p.q_bs = p.k_bs = p.v_bs = (long)N * 3 * Da; p.q_rs = p.k_rs = p.v_rs = 3 * Da; p.q_hs = p.k_hs = p.v_hs = hd;
        p.o_bs = Da; p.o_rs = Da;                         // one output row per image, compact [Bt][Da]
        p.B = Bt; p.H = d.heads; p.Nq = 1; p.Nk = N; p.scale = scale; p.dtype = d.adt; p.hd = hd;
        CK(vt_wrap(vt_attn_launch(p, s), "dino attention (CLS query)")); }
      { VtGemmParams p = lin(ws + w.att, d.adt, Da, L.proj_w, d.cdt, Da, L.proj_b, tok, VT_F32, tstride, Bt, D, Da, VT_ACT_NONE);
        p.colscale = L.ls1; p.residual = tok; p.ldr = tstride;
        CK(vt_wrap(vt_gemm_launch(p, s), "dino proj (CLS rows)")); }
      CK(vt_k_rownorm(tok, VT_F32, tstride, ws + w.xn, d.adt, D, L.ln2_w, L.ln2_b, Bt, D, d.eps, VT_NORM_LAYER, s));
      CK(ffn(L, Bt, tstride));
      continue;
    }
    { VtGemmParams p = lin(ws + w.xn, d.adt, D, L.qkv_w, d.cdt, D, L.qkv_b, ws + w.qkv, d.adt, 3 * Da, M, 3 * Da, D, VT_ACT_NONE);
      CK(vt_wrap(vt_gemm_launch(p, s), "dino qkv")); }
    { VtAttnParams p;
      memset(&p, 0, sizeof(p));
      p.Q = ws + w.qkv; p.K = ws + w.qkv + (size_t)Da * a; p.V = ws + w.qkv + (size_t)2 * Da * a; p.O = ws + w.att;
      p.q_bs = p.k_bs = p.v_bs = (long)N * 3 * Da; p.q_rs = p.k_rs = p.v_rs = 3 * Da; p.q_hs = p.k_hs = p.v_hs = hd;
      p.o_bs = (long)N * Da; p.o_rs = Da;
      p.B = Bt; p.H = d.heads; p.Nq = N; p.Nk = N; p.scale = scale; p.dtype = d.adt; p.hd = hd;
      CK(vt_wrap(vt_attn_launch(p, s), "dino attention")); }
    { VtGemmParams p = lin(ws + w.att, d.adt, Da, L.proj_w, d.cdt, Da, L.proj_b, tok, VT_F32, D, M, D, Da, VT_ACT_NONE);
      p.colscale = L.ls1; p.residual = tok; p.ldr = D;
      CK(vt_wrap(vt_gemm_launch(p, s), "dino proj")); }
    CK(vt_k_rownorm(tok, VT_F32, D, ws + w.xn, d.adt, D, L.ln2_w, L.ln2_b, M, D, d.eps, VT_NORM_LAYER, s));
    CK(ffn(L, M, D));
  }
  // 3. final LayerNorm: on the CLS rows only -> pooler_output (DINOv2), or on every token -> last_hidden_state (SigLIP)
  if (d.out_all) CK(vt_k_rownorm(tok, VT_F32, D, out, VT_F32, D, h->lnf_w, h->lnf_b, M, D, d.eps, VT_NORM_LAYER, s, h->range_flag));
  else CK(vt_k_rownorm(tok, VT_F32, (long)N * D, out, VT_F32, D, h->lnf_w, h->lnf_b, Bt, D, d.eps, VT_NORM_LAYER, s, h->range_flag));
  return VT_OK;
}

// ======================================================================================= MLP chain / concat
int vt_mlp(const void* x, long ldx, int B, int n_layers, const int* dims, const void* const* W, const float* const* b, int act,
           void* y, int ydt, long ldy, int cdt, int adt, void* tmp, vt_stream_t stream) {
  if (!x || !dims || !W || !b || !y || n_layers < 1 || n_layers > 8) return vt_fail(VT_ERR_ARG, "vt_mlp: bad argument");
  hipStream_t s = (hipStream_t)stream;
  int mx = 0;
  for (int i = 1; i < n_layers; ++i) mx = dims[i] > mx ? dims[i] : mx;
  if (n_layers > 1 && !tmp) return vt_fail(VT_ERR_ARG, "vt_mlp: tmp required");
  const void* cur = x; long ld = ldx;
  for (int i = 0; i < n_layers; ++i) {
    const bool last = i == n_layers - 1;
    void* dst = last ? y : (char*)tmp + (size_t)(i & 1) * B * mx * es(adt);
    VtGemmParams p = lin(cur, adt, ld, W[i], cdt, dims[i], b[i], dst, last ? ydt : adt, last ? ldy : dims[i + 1], B, dims[i + 1], dims[i],
                         last ? VT_ACT_NONE : act);
    CK(vt_wrap(vt_gemm_launch(p, s), "vt_mlp layer"));
    cur = dst; ld = dims[i + 1];
  }
  return VT_OK;
}

int vt_concat_obs(const float* cls1, const float* cls2, int dv, const float* state, int sdim, const float* forces, int fdim,
                  void* out, int odt, long ldo, int B, vt_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, (size_t)B * ldo * es(odt), s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "memset");
  CK(vt_k_place_cols(cls1, VT_F32, dv, out, odt, ldo, 0, B, dv, s));
  CK(vt_k_place_cols(cls2, VT_F32, dv, out, odt, ldo, dv, B, dv, s));
  CK(vt_k_place_cols(state, VT_F32, sdim, out, odt, ldo, 2 * dv, B, sdim, s));
  if (forces && fdim > 0) CK(vt_k_place_cols(forces, VT_F32, fdim, out, odt, ldo, 2 * dv + sdim, B, fdim, s));
  return VT_OK;
}
