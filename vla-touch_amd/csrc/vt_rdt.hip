// vt_rdt.hip — host driver for the RDT diffusion transformer and its DPM-Solver++ sampling loop
// (replaces models/rdt/model.py:126-165, models/rdt/blocks.py:72-202 and models/rdt_runner.py:108-165,225-250).
//
// MI355X-first differences from the reference's execution (results identical up to rounding):
//   * the language / image conditions are constant over the denoise steps, so their cross-attention K (after
//     k_norm) and V projections are computed ONCE per chunk and cached in HBM ([B, L, 2, H, 64] per block,
//     ~0.5 GB per sample for RDT-1B: sized for 288 GB) — the reference re-projects 4 374 image tokens in every
//     block of every step (86 % of its FLOPs);
//   * ctrl-freq embedding and the adapted state token are computed once per chunk;
//   * the residual stream is kept in fp32 (the reference rounds it to bf16 after every add);
//   * the whole predict_action (adaptors, caches, n_steps x 28 blocks, solver updates) is one C call that only
//     enqueues kernels on the caller's stream.
//
// weight order (w = cdt, vectors fp32 unless noted):
//   0 t_w1 [D][256] 1 t_b1 2 t_w2 [D][D] 3 t_b2   4 f_w1 5 f_b1 6 f_w2 7 f_b2
//   8 x_pos [horizon+3][D] fp32   9 lang_pos [Lmax][D] adt   10 img_pos [Limg][D] adt
//   per block (21): norm1  qkv_w [3D][D] qkv_b  q_norm k_norm  proj_w proj_b  norm2  cq_w cq_b  ckv_w [2D][D] ckv_b
//                   cq_norm ck_norm  cproj_w cproj_b  norm3  fc1_w fc1_b fc2_w fc2_b
//   final (5): norm_final  ffc1_w ffc1_b  ffc2_w [out][D] ffc2_b
//   adaptors: lang (n_lang layers: w, b), img (n_img layers), state (n_state layers); first-layer K = token dims.
#include <math.h>
#include <string.h>
#include <new>
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_host.h"
#include "vt_prof.h"
#include "../../include/vlatouch.h"

#define CK(x) do { int _r = (x); if (_r) return _r; } while (0)
static int es(int dt) { return dt == VT_F32 ? 4 : 2; }
static bool is16(int dt) { return dt == VT_BF16 || dt == VT_F16; }
// element i of a buffer of type dt (VT_F32 / VT_BF16 / VT_F16) as float, and back; rnd16 = round to that type's grid
__device__ __forceinline__ float ldx(const void* p, long i, int dt) {
  return dt == VT_F32 ? ((const float*)p)[i] : dt == VT_F16 ? h2f(((const half_t*)p)[i]) : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ float rnd16(float v, int dt) { return dt == VT_F32 ? v : dt == VT_F16 ? h2f(f2h(v)) : bf2f(f2bf(v)); }
__device__ __forceinline__ void stx(void* p, long i, float v, int dt) {
  if (dt == VT_F32) ((float*)p)[i] = v; else if (dt == VT_F16) ((half_t*)p)[i] = f2h(v); else ((bf16_t*)p)[i] = f2bf(v);
}

namespace {
struct Blk {
  const float *norm1, *qkv_b, *qn, *kn, *proj_b, *norm2, *cq_b, *ckv_b, *cqn, *ckn, *cproj_b, *norm3, *fc1_b, *fc2_b;
  const void *qkv_w, *proj_w, *cq_w, *ckv_w, *cproj_w, *fc1_w, *fc2_w;
  // fragment-packed second copies of the per-denoise-step Linears (vt_rdt_set_packed; null = not packed): vt_gemm_pw.hip
  const void *qkv_wp, *proj_wp, *cq_wp, *cproj_wp, *fc1_wp, *fc2_wp;
};
struct Adaptor { int n; const void* w[4]; const float* b[4]; int kin; const void* wp[4]; };

VtGemmParams lin(const void* A, int adt, long lda, const void* W, int cdt, long ldw, const float* b, void* C, int odt, long ldc, int M, int N,
                 int K, int act, const void* Wp = nullptr) {
  VtGemmParams p;
  memset(&p, 0, sizeof(p));
  p.Wp = Wp;
  p.A = A; p.W = W; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc;
  p.bias = b; p.act = act; p.groups = 1; p.splitk = 1; p.a_dtype = adt; p.w_dtype = cdt; p.c_dtype = odt;
  return p;
}
}  // namespace

struct vt_rdt_s {
  vt_rdt_desc d;
  const void *t_w1, *t_w2, *f_w1, *f_w2;
  const float *t_b1, *t_b2, *f_b1, *f_b2, *x_pos;
  const void *lang_pos, *img_pos;
  Blk blk[64];
  const float *normf, *ffc1_b, *ffc2_b;
  const void *ffc1_w, *ffc2_w;
  const void *ffc1_wp, *ffc2_wp, *t_w1p, *t_w2p;      // + the small per-step Linears (timestep embedder, final projection; state adaptor: Adaptor::wp)
  int io_dt = VT_BF16;          // 16-bit modes: the grid the start noise (and, with state_f32 = 0, the solver state) is rounded to = the reference's dtype
  int state_f32 = 1;            // 16-bit modes: keep the solver state, the network's x0 output and the final projection in fp32 (vt_rdt_set_state_precision)
  unsigned* range_flag = nullptr; // device word of the range guard (vt_rdt_set_range_flag), null = none
  float score_bound[64];        // per block: upper bound of |q . k| * scale in its cross-attention (vt_rdt_set_score_bounds), 0 = unknown
  Adaptor lang, img, state;
};

int vt_rdt_num_weights(const vt_rdt_desc* d) { return 11 + 21 * d->depth + 5 + 2 * (d->n_lang + d->n_img + d->n_state); }

int vt_rdt_create(const vt_rdt_desc* desc, const void* const* w, int n, vt_rdt_t* out) {
  if (!desc || !w || !out) return vt_fail(VT_ERR_ARG, "vt_rdt_create: null argument");
  const vt_rdt_desc& d = *desc;
  if (d.hidden % 64 || d.hidden / d.heads != 64 || d.depth < 1 || d.depth > 64 || d.horizon < 1 || d.out_dim < 1)
    return vt_fail(VT_ERR_ARG, "vt_rdt_create: unsupported config (head_dim must be 64)");
  if (d.n_lang < 1 || d.n_lang > 4 || d.n_img < 1 || d.n_img > 4 || d.n_state < 1 || d.n_state > 4) return vt_fail(VT_ERR_ARG, "vt_rdt_create: adaptor depth 1..4");
  if (d.lang_dim % 16 || d.img_dim % 16 || (2 * d.state_dim) % 16) return vt_fail(VT_ERR_ARG, "vt_rdt_create: token dims must be multiples of 16");
  if (d.cdt != d.adt || (d.cdt != VT_F32 && !is16(d.cdt))) return vt_fail(VT_ERR_ARG, "vt_rdt_create: cdt == adt in {fp32, bf16, fp16}");
  if (n != vt_rdt_num_weights(desc)) return vt_fail(VT_ERR_ARG, "vt_rdt_create: expected %d weights, got %d", vt_rdt_num_weights(desc), n);
  for (int k = 0; k < n; ++k) if (!w[k]) return vt_fail(VT_ERR_ARG, "vt_rdt_create: weight %d is null", k);
  vt_rdt_s* h = new (std::nothrow) vt_rdt_s();
  if (!h) return vt_fail(-12, "out of host memory");
  h->d = d;
  h->io_dt = d.adt == VT_F16 ? VT_F16 : VT_BF16;
  int i = 0;
  auto F = [&]() { return (const float*)w[i++]; };
  h->t_w1 = w[i++]; h->t_b1 = F(); h->t_w2 = w[i++]; h->t_b2 = F();
  h->f_w1 = w[i++]; h->f_b1 = F(); h->f_w2 = w[i++]; h->f_b2 = F();
  h->x_pos = F(); h->lang_pos = w[i++]; h->img_pos = w[i++];
  for (int l = 0; l < d.depth; ++l) {
    Blk& b = h->blk[l];
    b.norm1 = F(); b.qkv_w = w[i++]; b.qkv_b = F(); b.qn = F(); b.kn = F(); b.proj_w = w[i++]; b.proj_b = F();
    b.norm2 = F(); b.cq_w = w[i++]; b.cq_b = F(); b.ckv_w = w[i++]; b.ckv_b = F(); b.cqn = F(); b.ckn = F(); b.cproj_w = w[i++]; b.cproj_b = F();
    b.norm3 = F(); b.fc1_w = w[i++]; b.fc1_b = F(); b.fc2_w = w[i++]; b.fc2_b = F();
  }
  h->normf = F(); h->ffc1_w = w[i++]; h->ffc1_b = F(); h->ffc2_w = w[i++]; h->ffc2_b = F();
  auto rd = [&](Adaptor& a, int nl, int kin) { a.n = nl; a.kin = kin; for (int k = 0; k < nl; ++k) { a.w[k] = w[i++]; a.b[k] = F(); } };
  rd(h->lang, d.n_lang, d.lang_dim); rd(h->img, d.n_img, d.img_dim); rd(h->state, d.n_state, 2 * d.state_dim);
  *out = h;
  return VT_OK;
}
void vt_rdt_destroy(vt_rdt_t h) { delete h; }

// Per-block upper bounds of the scaled cross-attention scores.  cross_attn.q_norm / k_norm are per-head RMS norms (blocks.py:86-87, 112-113):
// in the mean-square form |q_normed|_2 <= 8 max|w_q| and |k_normed|_2 <= 8 max|w_k|, so |q . k| / 8 <= 8 max|w_q| max|w_k|: the caller (which
// holds the weights) passes that number per block and the cached cross-attention runs its softmax against it instead of a running maximum
// (vt_attn_kvt.hip).  0 = no bound (the variance form of timm <= 1.0.8 has none): online softmax.
int vt_rdt_set_score_bounds(vt_rdt_t h, const float* bounds, int n) {
  if (!h || !bounds || n != h->d.depth) return vt_fail(VT_ERR_ARG, "vt_rdt_set_score_bounds: one bound per block");
  for (int l = 0; l < n; ++l) h->score_bound[l] = bounds[l] > 0.f ? bounds[l] : 0.f;
  return VT_OK;
}

// Precision of what the DPM-Solver++ loop carries between network evaluations in the 16-bit mode.  fp32_state = 1 (default): the final projection
// writes fp32, the x0 predictions and the solver state stay fp32 (only the copy fed to the action-token adaptor is rounded, as any bf16 GEMM operand is) —
// strictly closer to the fp32 reference (at RDT-1B, B = 32: |chunk - oracle| 7.7e-3 -> see DESIGN.md section 3).  fp32_state = 0: the reference's own
// rounding points in bf16 (`noisy_action.to(dtype)` after every scheduler step, rdt_runner.py:160; bf16 model output).  No effect in fp32 mode.
// The reference's dtype when it differs from the engine's 16-bit compute type: a bf16 model evaluated with IEEE fp16 activations (adt = cdt = VT_F16:
// same width and MFMA rate, 3 more mantissa bits; the bf16 weights convert exactly) still draws / rounds its start noise — and, with the reference's
// rounding points, its solver state — on the bf16 grid.  io_dtype in {VT_BF16, VT_F16}.
int vt_rdt_set_io_dtype(vt_rdt_t h, int io_dtype) {
  if (!h || (io_dtype != VT_BF16 && io_dtype != VT_F16)) return vt_fail(VT_ERR_ARG, "vt_rdt_set_io_dtype: bf16 or fp16");
  h->io_dt = io_dtype;
  return VT_OK;
}
int vt_rdt_set_range_flag(vt_rdt_t h, unsigned* word) {
  if (!h) return vt_fail(VT_ERR_ARG, "vt_rdt_set_range_flag: null handle");
  h->range_flag = word;
  return VT_OK;
}
int vt_rdt_set_state_precision(vt_rdt_t h, int fp32_state) {
  if (!h) return vt_fail(VT_ERR_ARG, "vt_rdt_set_state_precision: null handle");
  h->state_f32 = fp32_state ? 1 : 0;
  return VT_OK;
}

// Fragment-packed second copies of the Linears of the denoise loop (qkv, proj, cross q, cross proj, fc1, fc2 of every block + the final
// fc1): the caller owns `buf` (vt_rdt_packed_bytes(h) bytes, resident as long as the handle is used); the packing kernels are enqueued
// on `stream`.  Returns 0 bytes when the configuration has no use for them (fp32 mode, hidden size not a multiple of 512).
static bool pk_ok(int N, int K) { return N % 64 == 0 && K % 256 == 0; }      // what vt_gemm_pws.hip / vt_gemm_pw.hip can take
size_t vt_rdt_packed_bytes(vt_rdt_t h) {
  if (!h || !is16(h->d.cdt) || h->d.hidden % 512) return 0;
  const size_t D = h->d.hidden, DD = D * D * 2;
  size_t n = (size_t)h->d.depth * 8 * DD + DD;                                // blocks + final fc1
  if (pk_ok(h->d.out_dim, (int)D)) n += (size_t)h->d.out_dim * D * 2;           // final fc2
  n += D * 256 * 2 + DD;                                                       // timestep embedder
  for (int i = 0; i < h->state.n; ++i) { const int K = i == 0 ? h->state.kin : (int)D; if (pk_ok((int)D, K)) n += D * K * 2; }
  return n;
}
int vt_rdt_set_packed(vt_rdt_t h, void* buf, vt_stream_t stream) {
  if (!h) return vt_fail(VT_ERR_ARG, "vt_rdt_set_packed: null handle");
  if (!vt_rdt_packed_bytes(h)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_rdt_set_packed: this configuration has no packed weights");
  if (!buf) return vt_fail(VT_ERR_ARG, "vt_rdt_set_packed: null buffer");
  const int D = h->d.hidden;
  char* o = (char*)buf;
  auto pk = [&](const void* W, int N, int K, const void** slot) -> int {
    if (!pk_ok(N, K)) { *slot = nullptr; return VT_OK; }
    const int r = vt_pack_w32(W, K, o, N, K, stream);
    if (r) return r;
    *slot = o; o += (size_t)N * K * 2;
    return VT_OK;
  };
  for (int l = 0; l < h->d.depth; ++l) {
    Blk& b = h->blk[l];
    CK(pk(b.qkv_w, 3 * D, D, &b.qkv_wp)); CK(pk(b.proj_w, D, D, &b.proj_wp)); CK(pk(b.cq_w, D, D, &b.cq_wp)); CK(pk(b.cproj_w, D, D, &b.cproj_wp));
    CK(pk(b.fc1_w, D, D, &b.fc1_wp)); CK(pk(b.fc2_w, D, D, &b.fc2_wp));
  }
  CK(pk(h->ffc1_w, D, D, &h->ffc1_wp));
  CK(pk(h->ffc2_w, h->d.out_dim, D, &h->ffc2_wp));
  CK(pk(h->t_w1, D, 256, &h->t_w1p)); CK(pk(h->t_w2, D, D, &h->t_w2p));
  for (int i = 0; i < h->state.n; ++i) CK(pk(h->state.w[i], D, i == 0 ? h->state.kin : D, &h->state.wp[i]));
  return VT_OK;
}

namespace {
inline int lpad64(int L) { return (L + 63) / 64 * 64; }
constexpr int RDT_MAX_SPLITK = 16;
constexpr int RDT_SK_CNT = 4096;
struct RWs {
  size_t lang_c, img_c, tmpA, tmpB, state_tok, freq_emb, t_emb, emb_tmp, sin, kv_lang, kv_img, x, xn, qkv, q, att, hid, sa_in, sa_tmpA, sa_tmpB,
      out_tok, x0_cur, x0_prev, noisy, noisy_a, slab, sk_cnt, rs_part, kv_small, total;
  size_t kv_lang_blk, kv_img_blk;   // bytes per block
  size_t slab_bytes;                // split-K scratch of the small-batch Linears (0 when M is large enough without it)
  size_t attn_part; int attn_parts; // key-range parts of the cached cross-attention at small batch (1 = off)
};
// the fused K | V projection of a condition with R rows, as cache_cond launches it (pointers left null: only shapes and dtypes decide the path)
VtGemmParams cond_kv_params(const vt_rdt_desc& d, int R) {
  VtGemmParams p = lin(nullptr, d.adt, d.hidden, nullptr, d.cdt, d.hidden, nullptr, nullptr, d.adt, d.hidden, R, 2 * d.hidden, d.hidden, VT_ACT_NONE);
  p.cmap = 3; p.cmap_T = lpad64(R) / 64;
  return p;
}
RWs rcarve(const vt_rdt_s* h, int B, int L) {
  const vt_rdt_desc& d = h->d;
  const int a = es(d.adt), D = d.hidden, N = d.horizon + 3, Li = d.img_len;
  RWs w; size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += (b + 255) / 256 * 256; return r; };
  w.lang_c = take((size_t)B * L * D * a);
  w.img_c = take((size_t)B * Li * D * a);
  const size_t rows = (size_t)B * (Li > L ? Li : L);
  w.tmpA = take(rows * D * a); w.tmpB = take(rows * D * a);
  // row-major [R][2D] scratch of the K | V product of a condition that falls off the large-GEMM path (few rows: the language tokens at batch 1..3,
  // every condition of the tiny test configs) — its own buffer: 2 R D elements do not fit tmpA when 2 R > max(Li, L) * B
  { size_t rs = 0;
    if (is16(d.adt)) for (int R : {B * L, B * Li}) if (!vt_gemm_can_fuse_headnorm(cond_kv_params(d, R)) && (size_t)R > rs) rs = (size_t)R;
    w.kv_small = take(rs * 2 * D * a); }
  w.state_tok = take((size_t)B * D * a); w.freq_emb = take((size_t)B * D * a); w.t_emb = take((size_t)B * D * a);
  w.emb_tmp = take((size_t)B * D * a); w.sin = take((size_t)B * 256 * a);
  const int n_lang_blk = (d.depth + 1) / 2, n_img_blk = d.depth / 2;
  // per block: fp32 mode [B*L][2D] (K | V interleaved per token); bf16 mode K [B*L][D] followed by Vt [B][H][64][Lpad]
  // bf16: per-head tile stream of 16-KiB [K | Vt] tiles over the B*L rows (vt_attn_kvt.hip); fp32: row-major [B*L][2D]
  w.kv_lang_blk = ((size_t)D * 2 * lpad64(B * L) * a + 255) / 256 * 256;
  w.kv_img_blk = ((size_t)D * 2 * lpad64(B * Li) * a + 255) / 256 * 256;
  w.kv_lang = take(w.kv_lang_blk * n_lang_blk);
  w.kv_img = take(w.kv_img_blk * n_img_blk);
  const size_t M = (size_t)B * N;
  w.rs_part = take(M * (2 * (D / 128) + 4) * 8);      // fused RMSNorm hand-off: (sum of squares, sum) per (row, 64 columns)
  w.x = take(M * D * 4); w.xn = take(M * D * a); w.qkv = take(M * 3 * D * a); w.q = take(M * D * a); w.att = take(M * D * a); w.hid = take(M * D * a);
  w.sa_in = take((size_t)B * d.horizon * 2 * d.state_dim * a);
  w.sa_tmpA = take((size_t)B * d.horizon * D * a); w.sa_tmpB = take((size_t)B * d.horizon * D * a);
  w.out_tok = take(M * d.out_dim * 4);                 // (sized for the fp32 form of vt_rdt_sample's solver loop)
  w.x0_cur = take((size_t)B * d.horizon * d.out_dim * 4); w.x0_prev = take((size_t)B * d.horizon * d.out_dim * 4);
  w.noisy = take((size_t)B * d.horizon * d.out_dim * 4); w.noisy_a = take((size_t)B * d.horizon * d.out_dim * a);
  w.slab_bytes = M <= 512 ? (size_t)RDT_MAX_SPLITK * M * 3 * D * 4 : 0;
  w.slab = take(w.slab_bytes);
  w.sk_cnt = take(RDT_SK_CNT * sizeof(int));    // ticket counters of the small-M tile (vt_gemm_pws.hip), zeroed once per call
  // cached cross-attention: B*H blocks stream a sample's whole key range each; below ~512 blocks split the range (flash-decoding)
  w.attn_parts = 1;
  { const int bh = B * d.heads; if (bh < 512 && N <= 128) { w.attn_parts = 512 / bh; if (w.attn_parts > 16) w.attn_parts = 16; if (w.attn_parts < 1) w.attn_parts = 1; } }
  w.attn_part = take(vt_attn_kvt_part_bytes(B, d.heads, N, w.attn_parts));
  w.total = o;
  return w;
}

struct RCtx { const vt_rdt_s* h; int B, L; char* ws; RWs w; hipStream_t s; int a;
              const void* pf_next = nullptr; size_t pf_bytes = 0;   // prefetch hint for the next rgemm on the weights-in-registers tile: the weights of the Linear after it
              bool out_f32 = false;       // the final projection writes out_tok in fp32 (vt_rdt_sample with the fp32 solver state)
              bool fuse_norm = false;     // residual Linears hand the RMSNorm that follows them to the next Linear (vt_gemm.h, xn_out / rs_part)
              bool rs_pending = false; }; // c.w.xn holds x * gain, un-normalised: the next Linear applies rstd from c.w.rs_part

// request the fused per-head RMSNorm epilogue when this GEMM takes the large-GEMM path; returns false -> caller runs vt_k_headnorm
bool fuse_headnorm(VtGemmParams& p, const float* w0, int c0_end, const float* w1, int c1_end, int mode) {
  if (!vt_gemm_can_fuse_headnorm(p)) return false;
  p.hn_w0 = w0; p.hn_c0_end = c0_end; p.hn_w1 = w1; p.hn_c1_end = c1_end; p.hn_eps = 1e-6f; p.hn_mode = mode;
  return true;
}

// Linear of the denoise loop.  At small batch (M = B*67 rows below ~512) M x N gives the generic kernel only a few dozen tiles, each
// walking the whole K = 2048 alone (36 us for 8 MB of weights); the k range is then split over up to 16 blocks per tile into fp32
// slabs and a second tiny kernel sums them and applies the Linear's epilogue.
// hn_*: head norm to apply when the GEMM did NOT fuse it (p.hn_w0 unset): folded into the slab reduction on the split path, else
// the caller's vt_k_headnorm kernels run (returns *hn_done = false).
// next_norm / xn_done: for a residual Linear into the fp32 stream (C == residual == x, full row width), the RMSNorm that follows it
// is computed by the slab reduction too (*xn_done = true: c.w.xn holds norm(x) * next_norm).
int rgemm(RCtx& c, VtGemmParams p, const char* what, const float* hn_w0 = nullptr, int hn_c0 = 0, const float* hn_w1 = nullptr, int hn_c1 = 0,
          bool* hn_done = nullptr, const float* next_norm = nullptr, bool* xn_done = nullptr, bool pw_fuse = true) {
  if (hn_done) *hn_done = false;
  if (xn_done) *xn_done = false;
  p.range_flag = c.h->range_flag;
  // (frozen, fragment-packed weights at small M: the split GEMM below runs on vt_gemm_pws.hip in its slab mode — p.Wp travels with q)
  const long tiles64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  const int nk = p.K / 64;
  // slices per tile on the fragment-packed small-M kernel: at most 4 (8 slices of a K = 2048 Linear finish the GEMM 1 us sooner and cost the
  // slab reduction 2 us more: batch 1 21.6 -> 21.0 ms; VLATOUCH_RDT_SPLIT_CAP for A/B)
  static const int split_cap = [] { const char* e = getenv("VLATOUCH_RDT_SPLIT_CAP"); return e && atoi(e) > 0 ? atoi(e) : 4; }();
  int S = (int)(512 / (tiles64 > 0 ? tiles64 : 1));
  if (S > RDT_MAX_SPLITK) S = RDT_MAX_SPLITK;
  if (S > nk / 2) S = nk / 2;
  if (p.Wp && S >= 2) {              // vt_gemm_pws.hip (slab mode) wants a power of two that leaves whole 4-k-tile chunks per slice
    int P = 1;
    while (P * 2 <= S && P * 2 <= split_cap && nk % (P * 2 * 4) == 0) P *= 2;
    S = P;
  }
  const bool small = c.w.slab_bytes > 0 && !vt_gemm_fast_eligible(p) && p.M <= 512 && p.K >= 512 && (p.K % 64) == 0 && (p.N % 4) == 0 && !p.hn_w0 &&
                     !p.hn_w1 && p.groups == 1 && p.taps == 0 && S >= 2 && (size_t)S * p.M * p.N * 4 <= c.w.slab_bytes;
  if (!small) {
    // producer side of the RMSNorm hand-off: decided on THIS launch's parameters (the per-run switch c.fuse_norm only says that block 0 qualifies):
    // a Linear that would not take the weights-in-registers tile with the hand-off fields set keeps the plain path, and the caller's row-norm kernel runs
    if (c.fuse_norm && pw_fuse && next_norm && xn_done && p.residual == p.C && p.c_dtype == VT_F32 && p.N == c.h->d.hidden && p.act == VT_ACT_NONE && p.ldc == p.N) {
      VtGemmParams q = p;
      q.xn_out = c.ws + c.w.xn; q.xn_ld = p.N; q.xn_gain = next_norm; q.xn_part = (float*)(c.ws + c.w.rs_part);
      q.rs_mode = c.h->d.rms_mode;      // the variance form hands over centred second moments (vt_gemm.h)
      if (vt_gemm_fast_eligible(q) && vt_gemm_pw_eligible(q)) {
        p = q;
        *xn_done = true;
        c.rs_pending = true;
      }
    }
    // prefetch hint for the next launch's weights: measured SLOWER (round 5, EXPERIMENTS.md: one batch at a time 363 -> 359 chunks/s, batch 1 20.31 -> 20.59 ms)
    // — off unless VLATOUCH_RDT_PREFETCH=1
    static const bool pf_on = [] { const char* e = getenv("VLATOUCH_RDT_PREFETCH"); return e && atoi(e) != 0; }();
    if (pf_on && c.pf_next && p.Wp && vt_gemm_fast_eligible(p) && vt_gemm_pw_eligible(p)) { p.pf_ptr = c.pf_next; p.pf_bytes = c.pf_bytes; }
    c.pf_next = nullptr;
    return vt_wrap(vt_gemm_launch(p, c.s), what);
  }
  // split path (small batch): the slab reduction behind the GEMM carries the hint, on extra blocks of its own launch
  static const bool pf_small = [] { const char* e = getenv("VLATOUCH_RDT_PREFETCH"); return e && atoi(e) != 0; }();
  const void* pfp = pf_small ? c.pf_next : nullptr;
  const size_t pfb = c.pf_bytes;
  c.pf_next = nullptr;
  VtGemmParams q = p;
  q.C = c.ws + c.w.slab; q.c_dtype = VT_F32; q.ldc = p.N; q.splitk = S; q.c_slab = (long)p.M * p.N;
  q.bias = nullptr; q.act = VT_ACT_NONE; q.colscale = nullptr; q.residual = nullptr;
  CK(vt_wrap(vt_gemm_launch(q, c.s), what));
  if (next_norm && xn_done && p.residual == p.C && p.c_dtype == VT_F32 && p.N == c.h->d.hidden && p.N <= 2048 && p.act == VT_ACT_NONE &&
      !p.colscale && p.ldc == p.N && p.ldr == p.N) {
    *xn_done = true;
    return vt_wrap(vt_k_slab_reduce_norm((const float*)(c.ws + c.w.slab), S, q.c_slab, p.M, p.N, p.bias, (float*)p.C, p.ldc, next_norm, nullptr, 1e-6f,
                                         c.h->d.rms_mode, c.ws + c.w.xn, c.h->d.adt, p.N, c.s, pfp, pfb), what);
  }
  const bool hn = hn_w0 && (p.N % 64) == 0;
  if (hn && hn_done) *hn_done = true;
  return vt_wrap(vt_k_slab_reduce((const float*)(c.ws + c.w.slab), S, q.c_slab, p.M, p.N, p.bias, p.act, p.colscale, p.residual, p.ldr, p.C, p.c_dtype,
                                  p.ldc, hn ? hn_w0 : nullptr, hn ? hn_w1 : nullptr, hn_c0, hn_c1, 1e-6f, c.h->d.rms_mode, c.s, pfp, pfb), what);
}

// adaptor MLP: Linear (gelu_tanh Linear)*  — final layer writes `dst` (ld = D) with optional per-row-in-sample residual (pos embed)
int run_adaptor(RCtx& c, const Adaptor& ad, const void* in, long rows_per_sample, int samples, void* dst, const void* pos, char* tA, char* tB) {
  const vt_rdt_desc& d = c.h->d;
  const int D = d.hidden;
  const void* cur = in; long ld = ad.kin; int K = ad.kin;
  for (int i = 0; i < ad.n; ++i) {
    const bool last = i == ad.n - 1;
    void* o = last ? dst : (void*)((i & 1) ? tB : tA);
    // GELU(tanh) sits BEFORE every Linear except the first (rdt_runner.py:97-101): fuse it into the previous epilogue
    VtGemmParams p = lin(cur, d.adt, ld, ad.w[i], d.cdt, K, ad.b[i], o, d.adt, D, (int)(rows_per_sample * samples), D, K, last ? VT_ACT_NONE : VT_ACT_GELU_TANH,
                         ad.wp[i]);
    if (last && pos) {   // + position embedding, broadcast over samples: one group per sample
      p.M = (int)rows_per_sample; p.groups = samples; p.a_gs = rows_per_sample * ld; p.c_gs = rows_per_sample * D;
      p.residual = pos; p.ldr = D; p.r_gs = 0;
      CK(vt_wrap(vt_gemm_launch(p, c.s), "rdt adaptor"));
    } else {
      CK(rgemm(c, p, "rdt adaptor"));      // few rows (the state adaptor inside the denoise loop): the packed-weight small-M tile
    }
    cur = o; ld = D; K = D;
  }
  return VT_OK;
}

// TimestepEmbedder (blocks.py:28-66): sinusoid(cos|sin) -> Linear -> SiLU -> Linear ; t_dev[B] or scalar
int embed(RCtx& c, const float* t_dev, float t_host, const void* w1, const float* b1, const void* w2, const float* b2, size_t out_off,
          const void* w1p = nullptr, const void* w2p = nullptr) {
  const vt_rdt_desc& d = c.h->d;
  const int D = d.hidden;
  CK(vt_k_sinusoid(t_dev, t_host, c.ws + c.w.sin, d.adt, c.B, 256, 1, 0, 1, c.s));
  { VtGemmParams p = lin(c.ws + c.w.sin, d.adt, 256, w1, d.cdt, 256, b1, c.ws + c.w.emb_tmp, d.adt, D, c.B, D, 256, VT_ACT_SILU, w1p);
    CK(rgemm(c, p, "rdt embed 1")); }
  { VtGemmParams p = lin(c.ws + c.w.emb_tmp, d.adt, D, w2, d.cdt, D, b2, c.ws + out_off, d.adt, D, c.B, D, D, VT_ACT_NONE, w2p);
    CK(rgemm(c, p, "rdt embed 2")); }
  return VT_OK;
}

int attn(RCtx& c, const void* Q, long q_rs, const void* K, const void* V, long kv_rs, int Nq, int Nk, const uint8_t* mask, void* O) {
  const vt_rdt_desc& d = c.h->d;
  VtAttnParams p;
  memset(&p, 0, sizeof(p));
  p.Q = Q; p.K = K; p.V = V; p.O = O;
  p.q_bs = (long)Nq * q_rs; p.q_rs = q_rs; p.q_hs = 64;
  p.k_bs = p.v_bs = (long)Nk * kv_rs; p.k_rs = p.v_rs = kv_rs; p.k_hs = p.v_hs = 64;
  p.o_bs = (long)Nq * d.hidden; p.o_rs = d.hidden;
  p.kmask = mask; p.km_bs = Nk;
  p.B = c.B; p.H = d.heads; p.Nq = Nq; p.Nk = Nk; p.scale = 0.125f; p.dtype = d.adt;
  return vt_wrap(vt_attn_launch(p, c.s), "rdt attention");
}

// condition K/V caches for every block (cross_attn.kv + k_norm, blocks.py:107-109) from adapted conditions (+pos already added)
int cache_cond(RCtx& c) {
  const vt_rdt_desc& d = c.h->d;
  const int D = d.hidden;
  for (int l = 0; l < d.depth; ++l) {
    const Blk& b = c.h->blk[l];
    const bool lang = (l % 2) == 0;
    const int Lc = lang ? c.L : d.img_len;
    char* kv = lang ? c.ws + c.w.kv_lang + (size_t)(l / 2) * c.w.kv_lang_blk : c.ws + c.w.kv_img + (size_t)(l / 2) * c.w.kv_img_blk;
    const void* src = lang ? c.ws + c.w.lang_c : c.ws + c.w.img_c;
    if (is16(d.adt)) {
      // 16-bit: K (k_norm fused) and V go straight from the GEMM epilogues into the tile stream when the projection takes the
      // large-GEMM path; small shapes go row-major through tmpA / tmpB and the retile kernels.
      const int T = lpad64(c.B * Lc) / 64;
      // one launch per layer: K | V fused (N = 2D), both halves of the tile stream written from ONE pass over the condition rows
      VtGemmParams pkv = cond_kv_params(d, c.B * Lc);
      pkv.A = src; pkv.W = b.ckv_w; pkv.bias = b.ckv_b; pkv.C = kv;
      if (fuse_headnorm(pkv, b.ckn, D, nullptr, D, d.rms_mode)) {
        CK(vt_wrap(vt_gemm_launch(pkv, c.s), "rdt cond kv"));
      } else {
        // small condition (the language tokens below 128 rows: batch 1..3): ONE K | V product (N = 2D) on the split-K path of the denoise loop's
        // Linears, k_norm folded into its slab reduction, row-major [rows][2D] into its own scratch (kv_small, sized by rcarve with the same test); then one retile launch.  (Two 32-row GEMMs walking
        // K = 2048 on 32 blocks each + head norm + two retile launches were 75 us per layer at batch 1; this is 20.)
        const int rows = c.B * Lc;
        char* kvs = c.ws + c.w.kv_small;
        VtGemmParams pk = lin(src, d.adt, D, b.ckv_w, d.cdt, D, b.ckv_b, kvs, d.adt, 2 * D, rows, 2 * D, D, VT_ACT_NONE);
        bool folded = false;
        CK(rgemm(c, pk, "rdt cond kv (small)", b.ckn, D, nullptr, D, &folded));
        if (!folded) CK(vt_k_headnorm(kvs, d.adt, 2 * D, d.heads, (long)rows, b.ckn, 1e-6f, d.rms_mode, c.s));
        CK(vt_wrap(vt_k_retile_kv(kvs, kvs + (size_t)D * c.a, 2 * D, kv, rows, T, d.heads, c.s), "rdt cond retile"));
      }
    } else {
      VtGemmParams p = lin(src, d.adt, D, b.ckv_w, d.cdt, D, b.ckv_b, kv, d.adt, 2 * D, c.B * Lc, 2 * D, D, VT_ACT_NONE);
      CK(vt_wrap(vt_gemm_launch(p, c.s), "rdt cond kv"));
      CK(vt_k_headnorm(kv, d.adt, 2 * D, d.heads, (long)c.B * Lc, b.ckn, 1e-6f, d.rms_mode, c.s));
    }
  }
  return VT_OK;
}

// cross-attention of N query rows per sample against block l's cached condition
int cross_attn(RCtx& c, int l, const uint8_t* lang_mask, int N) {
  const vt_rdt_desc& d = c.h->d;
  const int D = d.hidden, a = c.a;
  const bool lang = (l % 2) == 0;
  const int Lc = lang ? c.L : d.img_len;
  const char* kv = lang ? c.ws + c.w.kv_lang + (size_t)(l / 2) * c.w.kv_lang_blk : c.ws + c.w.kv_img + (size_t)(l / 2) * c.w.kv_img_blk;
  if (is16(d.adt)) {
    VtAttnKvtParams p;
    memset(&p, 0, sizeof(p));
    p.Q = c.ws + c.w.q; p.KV = kv; p.O = c.ws + c.w.att;
    p.q_bs = (long)N * D; p.q_rs = D; p.o_bs = (long)N * D; p.o_rs = D;
    p.kmask = lang ? lang_mask : nullptr;
    p.B = c.B; p.H = d.heads; p.Nq = N; p.Nk = Lc; p.T = lpad64(c.B * Lc) / 64; p.scale = 0.125f;
    p.fixed_max = c.h->score_bound[l];
    p.dtype = d.adt;
    p.range_flag = c.h->range_flag;
    if (c.w.attn_parts > 1 && Lc >= 64 * 2 * c.w.attn_parts) { p.parts = c.w.attn_parts; p.part_ws = (float*)(c.ws + c.w.attn_part); }
    return vt_wrap(vt_attn_kvt_launch(p, c.s), "rdt cross attention (cached K / Vt)");
  }
  return attn(c, c.ws + c.w.q, D, kv, kv + (size_t)D * a, 2 * D, N, Lc, lang ? lang_mask : nullptr, c.ws + c.w.att);
}

// the Linear about to run reads c.w.xn: if that is the un-normalised x * gain of a fused RMSNorm hand-off, it applies the rows' rstd itself — provided it
// takes the weights-in-registers tile (the only kernel with the consumer side); otherwise the row-norm kernel rebuilds xn from the fp32 stream
// (complete: the producer wrote it) with the same gain, and the Linear runs plain.  `gain` = the norm weight the producer was given.
int take_rstd(RCtx& c, VtGemmParams& p, const float* gain) {
  if (!c.rs_pending) return VT_OK;
  c.rs_pending = false;
  VtGemmParams q = p;
  q.rs_part = (const float*)(c.ws + c.w.rs_part); q.rs_n = 2 * (c.h->d.hidden / 128); q.rs_inv_k = 1.0f / (float)c.h->d.hidden; q.rs_eps = 1e-6f;
  q.rs_mode = c.h->d.rms_mode;
  if (q.c_dtype != VT_F32 && vt_gemm_fast_eligible(q) && vt_gemm_pw_eligible(q)) { p = q; return VT_OK; }
  const vt_rdt_desc& d = c.h->d;
  return vt_k_rownorm((const float*)(c.ws + c.w.x), VT_F32, d.hidden, c.ws + c.w.xn, d.adt, d.hidden, gain, nullptr, p.M, d.hidden, 1e-6f, d.rms_mode, c.s);
}

// blocks + final layer on the fp32 stream x [B*(horizon+3)][D]; writes out_tok [B*(horizon+3)][out_dim] (adt)
int run_blocks(RCtx& c, const uint8_t* lang_mask) {
  const vt_rdt_desc& d = c.h->d;
  const int D = d.hidden, N = d.horizon + 3, M = c.B * N, a = c.a;
  float* x = (float*)(c.ws + c.w.x);
  bool xn_ready = false;              // c.w.xn already holds the next norm of x (fused into the previous residual Linear's slab reduction)
  // RMSNorm hand-off (VtGemmParams::xn_out / rs_part): when every Linear of a block runs on the weights-in-registers tile (batch 32: M = 2144) the three
  // norm launches of a block disappear — the residual Linear writes x * gain + sums of squares, the next Linear scales its rows by rstd.
  // Both RmsNorm forms (round 5: the producer hands over the row sums beside the sums of squares, which is what the variance form of
  // timm <= 1.0.8 needs); VLATOUCH_RDT_FUSE_NORM=0 for A/B.
  {
    static const bool on = [] { const char* e = getenv("VLATOUCH_RDT_FUSE_NORM"); return !e || atoi(e) != 0; }();
    const Blk& b0 = c.h->blk[0];
    VtGemmParams pr = lin(c.ws + c.w.att, d.adt, D, b0.proj_w, d.cdt, D, b0.proj_b, x, VT_F32, D, M, D, D, VT_ACT_NONE, b0.proj_wp);
    pr.residual = x; pr.ldr = D;
    VtGemmParams c1 = lin(c.ws + c.w.xn, d.adt, D, b0.qkv_w, d.cdt, D, b0.qkv_b, c.ws + c.w.qkv, d.adt, 3 * D, M, 3 * D, D, VT_ACT_NONE, b0.qkv_wp);
    VtGemmParams c2 = lin(c.ws + c.w.xn, d.adt, D, b0.cq_w, d.cdt, D, b0.cq_b, c.ws + c.w.q, d.adt, D, M, D, D, VT_ACT_NONE, b0.cq_wp);
    c.fuse_norm = on && (d.rms_mode == VT_NORM_RMS_MEANSQ || d.rms_mode == VT_NORM_RMS_VAR) && c.w.slab_bytes == 0 && D % 128 == 0 && 2 * (D / 128) <= 32 && vt_gemm_fast_eligible(pr) &&
                  vt_gemm_pw_eligible(pr) && vt_gemm_fast_eligible(c1) && vt_gemm_pw_eligible(c1) && vt_gemm_fast_eligible(c2) && vt_gemm_pw_eligible(c2);
    c.rs_pending = false;
  }
  for (int l = 0; l < d.depth; ++l) {
    const Blk& b = c.h->blk[l];
    const float* norm_after = l + 1 < d.depth ? c.h->blk[l + 1].norm1 : c.h->normf;     // the norm that follows this block's fc2
    // --- self attention
    if (!xn_ready) CK(vt_k_rownorm(x, VT_F32, D, c.ws + c.w.xn, d.adt, D, b.norm1, nullptr, M, D, 1e-6f, d.rms_mode, c.s));
    xn_ready = false;
    { VtGemmParams p = lin(c.ws + c.w.xn, d.adt, D, b.qkv_w, d.cdt, D, b.qkv_b, c.ws + c.w.qkv, d.adt, 3 * D, M, 3 * D, D, VT_ACT_NONE, b.qkv_wp);
      CK(take_rstd(c, p, b.norm1));
      bool fused = fuse_headnorm(p, b.qn, D, b.kn, 2 * D, d.rms_mode);   // q_norm | k_norm | (v untouched)
      bool folded = false;
      c.pf_next = b.proj_wp; c.pf_bytes = (size_t)D * D * a;
      CK(rgemm(c, p, "rdt qkv", fused ? nullptr : b.qn, D, b.kn, 2 * D, &folded));
      fused = fused || folded;
      if (!fused) {
        CK(vt_k_headnorm(c.ws + c.w.qkv, d.adt, 3 * D, d.heads, M, b.qn, 1e-6f, d.rms_mode, c.s));
        CK(vt_k_headnorm(c.ws + c.w.qkv + (size_t)D * a, d.adt, 3 * D, d.heads, M, b.kn, 1e-6f, d.rms_mode, c.s));
      } }
    CK(attn(c, c.ws + c.w.qkv, 3 * D, c.ws + c.w.qkv + (size_t)D * a, c.ws + c.w.qkv + (size_t)2 * D * a, 3 * D, N, N, nullptr, c.ws + c.w.att));
    { VtGemmParams p = lin(c.ws + c.w.att, d.adt, D, b.proj_w, d.cdt, D, b.proj_b, x, VT_F32, D, M, D, D, VT_ACT_NONE, b.proj_wp);
      p.residual = x; p.ldr = D;
      c.pf_next = b.cq_wp; c.pf_bytes = (size_t)D * D * a;
      CK(rgemm(c, p, "rdt proj", nullptr, 0, nullptr, 0, nullptr, b.norm2, &xn_ready)); }
    // --- cross attention against the cached condition K/V
    if (!xn_ready) CK(vt_k_rownorm(x, VT_F32, D, c.ws + c.w.xn, d.adt, D, b.norm2, nullptr, M, D, 1e-6f, d.rms_mode, c.s));
    xn_ready = false;
    { VtGemmParams p = lin(c.ws + c.w.xn, d.adt, D, b.cq_w, d.cdt, D, b.cq_b, c.ws + c.w.q, d.adt, D, M, D, D, VT_ACT_NONE, b.cq_wp);
      CK(take_rstd(c, p, b.norm2));
      bool fused = fuse_headnorm(p, b.cqn, D, nullptr, D, d.rms_mode);
      bool folded = false;
      c.pf_next = b.cproj_wp; c.pf_bytes = (size_t)D * D * a;
      CK(rgemm(c, p, "rdt cross q", fused ? nullptr : b.cqn, D, nullptr, D, &folded));
      fused = fused || folded;
      if (!fused) CK(vt_k_headnorm(c.ws + c.w.q, d.adt, D, d.heads, M, b.cqn, 1e-6f, d.rms_mode, c.s)); }
    CK(cross_attn(c, l, lang_mask, N));
    { VtGemmParams p = lin(c.ws + c.w.att, d.adt, D, b.cproj_w, d.cdt, D, b.cproj_b, x, VT_F32, D, M, D, D, VT_ACT_NONE, b.cproj_wp);
      p.residual = x; p.ldr = D;
      c.pf_next = b.fc1_wp; c.pf_bytes = (size_t)D * D * a;
      CK(rgemm(c, p, "rdt cross proj", nullptr, 0, nullptr, 0, nullptr, b.norm3, &xn_ready)); }
    // --- FFN (hidden = D, tanh-GELU)
    if (!xn_ready) CK(vt_k_rownorm(x, VT_F32, D, c.ws + c.w.xn, d.adt, D, b.norm3, nullptr, M, D, 1e-6f, d.rms_mode, c.s));
    xn_ready = false;
    { VtGemmParams p = lin(c.ws + c.w.xn, d.adt, D, b.fc1_w, d.cdt, D, b.fc1_b, c.ws + c.w.hid, d.adt, D, M, D, D, VT_ACT_GELU_TANH, b.fc1_wp);
      CK(take_rstd(c, p, b.norm3));
      c.pf_next = b.fc2_wp; c.pf_bytes = (size_t)D * D * a;
      CK(rgemm(c, p, "rdt fc1")); }
    { VtGemmParams p = lin(c.ws + c.w.hid, d.adt, D, b.fc2_w, d.cdt, D, b.fc2_b, x, VT_F32, D, M, D, D, VT_ACT_NONE, b.fc2_wp);
      p.residual = x; p.ldr = D;
      if (l + 1 < d.depth) { c.pf_next = c.h->blk[l + 1].qkv_wp; c.pf_bytes = (size_t)3 * D * D * a; }
      else { c.pf_next = c.h->ffc1_wp; c.pf_bytes = (size_t)D * D * a; }
      CK(rgemm(c, p, "rdt fc2", nullptr, 0, nullptr, 0, nullptr, norm_after, &xn_ready, l + 1 < d.depth)); }   // the final layer's fc1 is not a consumer of the hand-off
  }
  if (!xn_ready) CK(vt_k_rownorm(x, VT_F32, D, c.ws + c.w.xn, d.adt, D, c.h->normf, nullptr, M, D, 1e-6f, d.rms_mode, c.s));
  { VtGemmParams p = lin(c.ws + c.w.xn, d.adt, D, c.h->ffc1_w, d.cdt, D, c.h->ffc1_b, c.ws + c.w.hid, d.adt, D, M, D, D, VT_ACT_GELU_TANH, c.h->ffc1_wp);
    CK(rgemm(c, p, "rdt final fc1")); }
  { VtGemmParams p = lin(c.ws + c.w.hid, d.adt, D, c.h->ffc2_w, d.cdt, D, c.h->ffc2_b, c.ws + c.w.out_tok, c.out_f32 ? VT_F32 : d.adt, d.out_dim, M, d.out_dim, D, VT_ACT_NONE,
                         c.h->ffc2_wp);
    CK(rgemm(c, p, "rdt final fc2")); }
  return VT_OK;
}

// x[b][0] = t_emb[b|0], x[b][1] = freq_emb[b], x[b][2] = state token, x[b][3..] = action tokens; + x_pos (model.py:141-148)
__global__ void assemble_x_kernel(float* __restrict__ x, const void* t_emb, int t_bcast, const void* f_emb, const void* state_tok, long state_bs,
                                  const void* act_tok, long act_bs, const float* __restrict__ pos, int B, int N, int D, int dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * N * D) return;
  const int dcol = (int)(i % D);
  const long r = i / D;
  const int n = (int)(r % N), b = (int)(r / N);
  const void* src; long off;
  if (n == 0) { src = t_emb; off = (long)(t_bcast ? 0 : b) * D + dcol; }
  else if (n == 1) { src = f_emb; off = (long)b * D + dcol; }
  else if (n == 2) { src = state_tok; off = (long)b * state_bs + dcol; }
  else { src = act_tok; off = (long)b * act_bs + (long)(n - 3) * D + dcol; }
  x[i] = ldx(src, off, dt) + pos[(long)n * D + dcol];
}

// out[b][l][:] = cond[b][l][:] + pos[l][:]   (model.py:150-152), all in the activation dtype
__global__ void add_pos_kernel(const void* cond, const void* pos, void* out, int B, int Lc, int D, int dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Lc * D) return;
  const long pl = i % ((long)Lc * D);
  stx(out, i, ldx(cond, i, dt) + ldx(pos, pl, dt), dt);
}

// dst = src rounded to the activation dtype's grid (bf16 mode: noisy_action lives in bf16, rdt_runner.py:137-139,160; fp32 mode: a plain copy) —
// the solver state's first value, taken from the caller's start noise in one kernel (no runtime copy kernel in the step)
__global__ void take_xinit_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int round_dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = rnd16(src[i], round_dt);
}

// sa_in[b][t] = cat(noisy[b][t] (adt copy of the fp32 master), mask[b])   (rdt_runner.py:148)
__global__ void build_sa_in_kernel(const float* __restrict__ noisy, const void* mask, void* out, int B, int Hh, int S, int dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Hh * 2 * S) return;
  const int c = (int)(i % (2 * S));
  const long r = i / (2 * S);
  const int b = (int)(r / Hh);
  float v;
  if (c < S) v = noisy[r * S + c];
  else v = ldx(mask, (long)b * S + c - S, dt);
  stx(out, i, v, dt);
}

// x0 = out_tok[:, -horizon:, :]  (model.py:164) gathered contiguous
__global__ void take_actions_kernel(const void* out_tok, void* x0, int B, int N, int Hh, int S, int is_16, int dt16, unsigned* range_flag) {      // is_16: both buffers 16-bit (a plain copy), else both fp32; dt16: which 16-bit type
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Hh * S) return;
  const int c = (int)(i % S);
  const long r = i / S;
  const int t = (int)(r % Hh), b = (int)(r / Hh);
  const long src = ((long)b * N + (N - Hh) + t) * S + c;
  if (is_16) {
    const uint16_t u = ((const uint16_t*)out_tok)[src];
    ((uint16_t*)x0)[i] = u;
    const uint16_t em = dt16 == VT_F16 ? 0x7C00 : 0x7F80;          // exponent all ones = inf / NaN
    if ((u & em) == em) vt_range_note(range_flag, VT_RANGE_NONFINITE);
  } else {
    const float v = ((const float*)out_tok)[src];
    ((float*)x0)[i] = v;
    if (vt_nonfinite(v)) vt_range_note(range_flag, VT_RANGE_NONFINITE);
  }
}

// noisy = round(a*noisy + b0*x0 + b1*x0_prev) ; last step: * mask   (rdt_runner.py:158-163).  x0_dt: storage type of the x0 buffers; round_dt: the
// reference's `noisy_action.to(dtype)` after every step (VT_F32 = none: the solver state is kept in fp32); mask_dt: storage type of the action mask
__global__ void dpm_update_kernel(float* __restrict__ noisy, const void* x0, const void* x0p, float a, float b0, float b1, const void* mask, int last,
                                  int B, int Hh, int S, int x0_dt, int round_dt, int mask_dt, int sample_pred, float alpha_s, float sigma_s, void* x0_store,
                                  float* __restrict__ out, unsigned* range_flag) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Hh * S) return;
  const int c = (int)(i % S);
  const int b = (int)(i / ((long)Hh * S));
  float m0 = ldx(x0, i, x0_dt);
  if (!sample_pred) m0 = (noisy[i] - sigma_s * m0) / alpha_s;          // epsilon prediction -> x0
  if (x0_store) stx(x0_store, i, m0, x0_dt);
  float v = a * noisy[i] + b0 * m0;
  if (x0p) v += b1 * ldx(x0p, i, x0_dt);
  v = rnd16(v, round_dt);                                           // `noisy_action.to(state_traj.dtype)` (rdt_runner.py:160)
  if (last) v = rnd16(v * ldx(mask, (long)b * S + c, mask_dt), round_dt);
  if (vt_nonfinite(v)) vt_range_note(range_flag, VT_RANGE_NONFINITE);     // range guard: an overflow anywhere upstream arrives here as inf / NaN
  noisy[i] = v;
  if (out) out[i] = v;                                              // the last step also writes the caller's buffer
}

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

int make_rctx(RCtx& c, vt_rdt_t h, int B, int L, void* ws, vt_stream_t s) {
  if (!h || !ws) return vt_fail(VT_ERR_ARG, "rdt: null handle/workspace");
  if (B < 1 || L < 1 || L > h->d.max_lang_len) return vt_fail(VT_ERR_ARG, "rdt: bad batch / language length %d (max %d)", L, h->d.max_lang_len);
  c.h = h; c.B = B; c.L = L; c.ws = (char*)ws; c.w = rcarve(h, B, L); c.s = (hipStream_t)s; c.a = es(h->d.adt);
  return VT_OK;
}
}  // namespace

size_t vt_rdt_workspace_bytes(vt_rdt_t h, int B, int L) { return h ? rcarve(h, B, L).total : 0; }

// RDT.forward (models/rdt/model.py:126-165): x [B][horizon+1][D] (already adapted state+action tokens), freq[B], t (scalar or [B]),
// lang_c [B][L][D], img_c [B][img_len][D] (adapted, WITHOUT position embeddings), lang_mask [B][L] or NULL -> out [B][horizon][out_dim] (adt)
int vt_rdt_forward(vt_rdt_t h, const void* x_tokens, const float* freq, const float* t_dev, float t_host, int t_is_scalar, const void* lang_c,
                   const void* img_c, const uint8_t* lang_mask, void* out, int B, int L, void* workspace, vt_stream_t stream) {
  RCtx c;
  CK(make_rctx(c, h, B, L, workspace, stream));
  if (hipMemsetAsync(c.ws + c.w.sk_cnt, 0, RDT_SK_CNT * sizeof(int), c.s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "ticket counters");
  const vt_rdt_desc& d = h->d;
  const int D = d.hidden, N = d.horizon + 3, dt = d.adt;
  // conditions + position embeddings (model.py:150-152)
  hipLaunchKernelGGL(add_pos_kernel, g1((long)B * L * D), dim3(256), 0, c.s, lang_c, h->lang_pos, (void*)(c.ws + c.w.lang_c), B, L, D, dt);
  hipLaunchKernelGGL(add_pos_kernel, g1((long)B * d.img_len * D), dim3(256), 0, c.s, img_c, h->img_pos, (void*)(c.ws + c.w.img_c), B, d.img_len, D, dt);
  CK(vt_check_launch());
  CK(cache_cond(c));
  CK(embed(c, t_is_scalar ? nullptr : t_dev, t_host, h->t_w1, h->t_b1, h->t_w2, h->t_b2, c.w.t_emb));
  CK(embed(c, freq, 0.f, h->f_w1, h->f_b1, h->f_w2, h->f_b2, c.w.freq_emb));
  // x_tokens is [B][horizon+1][D]: the state token, then the horizon action tokens
  const long xbs = (long)(d.horizon + 1) * D;
  hipLaunchKernelGGL(assemble_x_kernel, g1((long)B * N * D), dim3(256), 0, c.s, (float*)(c.ws + c.w.x), (const void*)(c.ws + c.w.t_emb), t_is_scalar,
                     (const void*)(c.ws + c.w.freq_emb), x_tokens, xbs, (const void*)((const char*)x_tokens + (size_t)D * c.a), xbs, h->x_pos, B, N, D, dt);
  CK(vt_check_launch());
  CK(run_blocks(c, lang_mask));
  hipLaunchKernelGGL(take_actions_kernel, g1((long)B * d.horizon * d.out_dim), dim3(256), 0, c.s, (const void*)(c.ws + c.w.out_tok), out, B, N, d.horizon,
                     d.out_dim, is16(dt) ? 1 : 0, dt, h->range_flag);
  return vt_check_launch();
}

// RDTRunner.predict_action (rdt_runner.py:225-250 + 122-165) with the restated DPM-Solver++(2M) update
// x <- a_k x + b0_k x0_k + b1_k x0_{k-1}  (coefficients from the host: vlatouch/dpm.py).
int vt_rdt_sample(vt_rdt_t h, const void* lang_tokens, const uint8_t* lang_mask, const void* img_tokens, const void* state_tokens,
                  const void* action_mask, const float* ctrl_freqs, const float* x_init, int n_steps, const float* timesteps,
                  const float* coef /* [n_steps][5] = a, b0, b1, alpha_s, sigma_s */, int sample_pred, int adapted, float* out, int B, int L,
                  void* workspace, vt_stream_t stream) {
  RCtx c;
  CK(make_rctx(c, h, B, L, workspace, stream));
  if (!lang_tokens || !img_tokens || !state_tokens || !action_mask || !ctrl_freqs || !x_init || !timesteps || !coef || !out || n_steps < 1)
    return vt_fail(VT_ERR_ARG, "vt_rdt_sample: null argument");
  const vt_rdt_desc& d = h->d;
  const int D = d.hidden, N = d.horizon + 3, Hh = d.horizon, S = d.state_dim, dt = d.adt, a = c.a;
  const bool bf = is16(dt);                       // a 16-bit mode (bf16 or IEEE fp16 activations)
  if (d.out_dim != S) return vt_fail(VT_ERR_ARG, "vt_rdt_sample: action_dim must equal state_token_dim");
  hipStream_t s = c.s;
  if (hipMemsetAsync(c.ws + c.w.sk_cnt, 0, RDT_SK_CNT * sizeof(int), s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "ticket counters");
  // ---- once per chunk: adaptors (+pos), condition K/V caches, ctrl-freq embedding, adapted state token
  if (!adapted) {   // predict_action: raw encoder tokens -> adaptors (rdt_runner.py:240-242)
    CK(run_adaptor(c, h->lang, lang_tokens, L, B, c.ws + c.w.lang_c, h->lang_pos, c.ws + c.w.tmpA, c.ws + c.w.tmpB));
    CK(run_adaptor(c, h->img, img_tokens, d.img_len, B, c.ws + c.w.img_c, h->img_pos, c.ws + c.w.tmpA, c.ws + c.w.tmpB));
    CK(vt_k_place_cols(state_tokens, d.adt, S, c.ws + c.w.sa_in, d.adt, 2 * S, 0, B, S, s));
    CK(vt_k_place_cols(action_mask, d.adt, S, c.ws + c.w.sa_in, d.adt, 2 * S, S, B, S, s));
    CK(run_adaptor(c, h->state, c.ws + c.w.sa_in, 1, B, c.ws + c.w.state_tok, nullptr, c.ws + c.w.sa_tmpA, c.ws + c.w.sa_tmpB));
  } else {          // conditional_sample: lang/img/state tokens are already [.., hidden] (rdt_runner.py:122-134)
    hipLaunchKernelGGL(add_pos_kernel, g1((long)B * L * D), dim3(256), 0, s, lang_tokens, h->lang_pos, (void*)(c.ws + c.w.lang_c), B, L, D, dt);
    hipLaunchKernelGGL(add_pos_kernel, g1((long)B * d.img_len * D), dim3(256), 0, s, img_tokens, h->img_pos, (void*)(c.ws + c.w.img_c), B, d.img_len, D, dt);
    CK(vt_check_launch());
    if (hipMemcpyAsync(c.ws + c.w.state_tok, state_tokens, (size_t)B * D * a, hipMemcpyDeviceToDevice, s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "state copy");
  }
  CK(cache_cond(c));
  CK(embed(c, ctrl_freqs, 0.f, h->f_w1, h->f_b1, h->f_w2, h->f_b2, c.w.freq_emb));
  const long n = (long)B * Hh * S;
  // the start noise on the grid of the I/O dtype (the reference draws it in its dtype, rdt_runner.py:137-139): bf16 unless the handle says otherwise
  hipLaunchKernelGGL(take_xinit_kernel, g1(n), dim3(256), 0, s, x_init, (float*)(c.ws + c.w.noisy), n, bf ? h->io_dt : VT_F32);
  CK(vt_check_launch());
  char* x0_cur = c.ws + c.w.x0_cur;
  char* x0_prev = c.ws + c.w.x0_prev;
  // 16-bit mode with the fp32 solver state (the default, vt_rdt_set_state_precision): fp32 final projection / x0 buffers, no per-step rounding of the state
  const int st32 = bf && h->state_f32;
  const int x0_dt = (bf && !st32) ? dt : VT_F32;                  // storage type of out_tok / the x0 buffers
  const int round_dt = (bf && !st32) ? h->io_dt : VT_F32;         // per-step rounding of the state (the reference's dtype)
  c.out_f32 = st32;
  for (int k = 0; k < n_steps; ++k) {
    hipLaunchKernelGGL(build_sa_in_kernel, g1((long)B * Hh * 2 * S), dim3(256), 0, s, (const float*)(c.ws + c.w.noisy), action_mask, (void*)(c.ws + c.w.sa_in), B, Hh, S, dt);
    CK(vt_check_launch());
    CK(run_adaptor(c, h->state, c.ws + c.w.sa_in, Hh, B, c.ws + c.w.sa_tmpA, nullptr, c.ws + c.w.tmpA, c.ws + c.w.tmpB));   // action tokens -> sa_tmpA
    CK(embed(c, nullptr, timesteps[k], h->t_w1, h->t_b1, h->t_w2, h->t_b2, c.w.t_emb, h->t_w1p, h->t_w2p));
    hipLaunchKernelGGL(assemble_x_kernel, g1((long)B * N * D), dim3(256), 0, s, (float*)(c.ws + c.w.x), (const void*)(c.ws + c.w.t_emb), 1,
                       (const void*)(c.ws + c.w.freq_emb), (const void*)(c.ws + c.w.state_tok), (long)D, (const void*)(c.ws + c.w.sa_tmpA), (long)Hh * D,
                       h->x_pos, B, N, D, dt);
    CK(vt_check_launch());
    CK(run_blocks(c, lang_mask));
    hipLaunchKernelGGL(take_actions_kernel, g1(n), dim3(256), 0, s, (const void*)(c.ws + c.w.out_tok), (void*)x0_cur, B, N, Hh, S, x0_dt != VT_F32 ? 1 : 0, dt, h->range_flag);
    CK(vt_check_launch());
    const float* cf = coef + 5 * k;
    const bool last = k == n_steps - 1;
    hipLaunchKernelGGL(dpm_update_kernel, g1(n), dim3(256), 0, s, (float*)(c.ws + c.w.noisy), (const void*)x0_cur, (const void*)(cf[2] != 0.f ? x0_prev : nullptr), cf[0], cf[1],
                       cf[2], action_mask, last ? 1 : 0, B, Hh, S, x0_dt, round_dt, dt, sample_pred, cf[3], cf[4], (void*)(sample_pred ? nullptr : x0_cur), last ? out : (float*)nullptr, h->range_flag);
    CK(vt_check_launch());
    char* t = x0_cur; x0_cur = x0_prev; x0_prev = t;
  }
  return VT_OK;
}
