// vt_unet.hip — host driver for the conditional 1-D U-Nets of the interpolant controller and the
// velocity-score SDE sampler (replaces bridge/networks/conditional_unet_1D.py:194-247 and
// bridge/bridge_model.py:334-387).  Activations are CHANNEL-LAST ([net][b*T + t][C]) so every Conv1d /
// ConvTranspose1d is an implicit GEMM over contiguous rows; `nets` (v_net, s_net) run in one grouped launch.
//
//   resblock:  conv0 (split-K slabs) -> GN+Mish+FiLM -> conv1 (slabs) -> [1x1 residual conv] -> GN+Mish+residual
//   skip connections are written straight into the second half of the concat buffer of the up level
//   that consumes them; the up-sampled tensor is written into the first half (no torch.cat copy).
//
// vt_unet_weight_order (all packed [nets][...]; "w" = cdt, everything else fp32):
//   0 step_w1 [4*dsed][dsed]   1 step_b1   2 step_w2 [dsed][4*dsed]   3 step_b2
//   4 film_w [F][dsed+cond]    5 film_b [F]        F = sum over resblocks of 2*Cout, resblock order below
//   per resblock (down l.0, down l.1 for each level; mid.0, mid.1; up i.0, up i.1), 10 entries:
//       c0_w [Cout][k*Cin_pad]  c0_b  gn0_g  gn0_b  c1_w [Cout][k*Cout]  c1_b  gn1_g  gn1_b  res_w [Cout][Cin_pad]|NULL  res_b|NULL
//   per down-sample (levels-1):  w [C][3*C] (tap-major)  b
//   per up-sample (levels-1):    w_even [C][2*C] (taps k=1,k=3)   w_odd [C][2*C] (taps k=0,k=2)   b
//   final:  fc_w [C0][k*C0]  fc_b  fgn_g  fgn_b  out_w [input_dim][C0]  out_b
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_host.h"
#include "vt_uconv.h"
#include "vt_unet_int.h"

namespace {

struct View { char* p; long ld; long gs; };   // element strides; p is a byte pointer

}  // namespace

static int es(int dt) { return dt == VT_BF16 ? 2 : 4; }

int vt_unet_num_weights(const vt_unet_desc* d) {
  const int L = d->n_levels;
  return 6 + (2 * L + 2 + 2 * (L - 1)) * 10 + (L - 1) * 2 + (L - 1) * 3 + 6;
}

int vt_unet_create(const vt_unet_desc* desc, const void* const* w, int n, vt_unet_t* out) {
  if (!desc || !w || !out) return vt_fail(VT_ERR_ARG, "vt_unet_create: null argument");
  const vt_unet_desc& d = *desc;
  if (d.nets < 1 || d.nets > 2 || d.n_levels < 2 || d.n_levels > 4 || d.input_pad % 16 || d.input_pad < d.input_dim)
    return vt_fail(VT_ERR_ARG, "vt_unet_create: bad descriptor");
  if (n != vt_unet_num_weights(desc)) return vt_fail(VT_ERR_ARG, "vt_unet_create: expected %d weight pointers, got %d", vt_unet_num_weights(desc), n);
  for (int l = 0; l < d.n_levels; ++l)
    if (d.dims[l] % (8 * d.n_groups) || d.dims[l] % 16) return vt_fail(VT_ERR_ARG, "vt_unet_create: dims must be multiples of 16 and of 8*n_groups");
  vt_unet_s* h = new (std::nothrow) vt_unet_s();
  if (!h) return vt_fail(-12, "out of host memory");
  h->d = d;
  int i = 0;
  h->step_w1 = w[i++]; h->step_b1 = (const float*)w[i++]; h->step_w2 = w[i++]; h->step_b2 = (const float*)w[i++];
  h->film_w = w[i++]; h->film_b = (const float*)w[i++];
  const int L = d.n_levels;
  int nrb = 0;
  long F = 0;
  int cmax = d.input_pad;
  auto add = [&](int cin, int cout) {
    ResBlk& r = h->rb[nrb++];
    r.cin = cin; r.cin_pad = (cin + 15) / 16 * 16; r.cout = cout;
    r.c0_w = w[i++]; r.c0_b = (const float*)w[i++]; r.g0 = (const float*)w[i++]; r.be0 = (const float*)w[i++];
    r.c1_w = w[i++]; r.c1_b = (const float*)w[i++]; r.g1 = (const float*)w[i++]; r.be1 = (const float*)w[i++];
    r.res_w = w[i++]; r.res_b = (const float*)w[i++];
    r.film_off = F; F += 2 * cout;
    if (cout > cmax) cmax = cout;
    if (r.cin_pad > cmax) cmax = r.cin_pad;
  };
  for (int l = 0; l < L; ++l) { add(l == 0 ? d.input_dim : d.dims[l - 1], d.dims[l]); add(d.dims[l], d.dims[l]); }
  add(d.dims[L - 1], d.dims[L - 1]); add(d.dims[L - 1], d.dims[L - 1]);
  for (int u = 0; u < L - 1; ++u) { const int din = d.dims[L - 2 - u], dout = d.dims[L - 1 - u]; add(2 * dout, din); add(din, din); }
  h->nrb = nrb; h->F = F; h->cmax = cmax;
  for (int l = 0; l < L - 1; ++l) { h->down_w[l] = w[i++]; h->down_b[l] = (const float*)w[i++]; }
  for (int u = 0; u < L - 1; ++u) { h->up_we[u] = w[i++]; h->up_wo[u] = w[i++]; h->up_b[u] = (const float*)w[i++]; }
  h->fc_w = w[i++]; h->fc_b = (const float*)w[i++]; h->fg = (const float*)w[i++]; h->fbe = (const float*)w[i++];
  h->out_w = w[i++]; h->out_b = (const float*)w[i++];
  vt_unet_fused_init_meta(h);
  *out = h;
  return VT_OK;
}

void vt_unet_destroy(vt_unet_t h) { delete h; }

namespace {

constexpr int MAX_SPLITK = 8;

struct UWs {   // workspace carve (byte offsets)
  size_t xp, sin, h1, gin, film, slabs, bufX, bufY, bufH, bufR, cat[4], vs_out, total;
  long slab_stride;  // elements per split-K slice
};

UWs carve(const vt_unet_s* h, int B, int T) {
  const vt_unet_desc& d = h->d;
  const int a = es(d.adt);
  const long M = (long)B * T;
  UWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
  w.xp = take((size_t)M * d.input_pad * a);
  w.sin = take((size_t)B * d.dsed * a);
  w.h1 = take((size_t)d.nets * B * 4 * d.dsed * a);
  w.gin = take((size_t)d.nets * B * (d.dsed + d.cond_dim) * a);
  w.film = take((size_t)d.nets * B * h->F * 4);
  w.slab_stride = (long)d.nets * M * h->cmax;
  w.slabs = take((size_t)MAX_SPLITK * w.slab_stride * 4);
  const size_t act = (size_t)d.nets * M * h->cmax * a;
  w.bufX = take(act); w.bufY = take(act); w.bufH = take(act); w.bufR = take(act);
  for (int u = 0; u < d.n_levels - 1; ++u) {
    const int lvl = d.n_levels - 1 - u;
    w.cat[u] = take((size_t)d.nets * (M >> lvl) * 2 * d.dims[lvl] * a);
  }
  w.vs_out = take((size_t)d.nets * M * d.input_dim * 4);
  w.total = o;
  return w;
}

int pick_splitk(long M, long N, int nets, int K, int bk) {
  const long tiles = ((M + 63) / 64) * ((N + 63) / 64) * nets;
  int s = (int)((384 + tiles - 1) / tiles);
  const int nk = (K + bk - 1) / bk;
  if (s > nk) s = nk;
  if (s > MAX_SPLITK) s = MAX_SPLITK;
  if (s < 1) s = 1;
  return s;
}

struct Ctx {
  const vt_unet_s* h; int B, T; char* ws; UWs w; hipStream_t s;
  int adt, cdt, a_es;
};

// conv (taps k, stride 1 or 2) into fp32 split-K slabs; returns number of slices through *nsl
int conv_slabs(Ctx& c, View x, int cin, const void* W, int cout, int taps, int tin, int tout, int stride, int off0, int* nsl) {
  VtGemmParams p;
  memset(&p, 0, sizeof(p));
  const vt_unet_desc& d = c.h->d;
  p.A = x.p; p.W = W; p.C = c.ws + c.w.slabs;
  p.M = c.B * tout; p.N = cout; p.K = taps * cin;
  p.lda = x.ld; p.ldw = p.K; p.ldc = cout;
  p.taps = taps; p.cin = cin; p.tout = tout; p.tin = tin; p.stride = stride; p.off0 = off0; p.tstep = 1;
  p.groups = d.nets; p.splitk = pick_splitk(p.M, p.N, d.nets, p.K, c.cdt == VT_F32 ? 32 : 64);
  p.a_gs = x.gs; p.w_gs = (long)cout * p.K; p.c_gs = (long)p.M * cout; p.c_slab = c.w.slab_stride;
  p.a_dtype = c.adt; p.w_dtype = c.cdt; p.c_dtype = VT_F32;
  *nsl = p.splitk;
  // splitk == 1 still writes the raw accumulator (no bias): the GN kernel adds the conv bias.
  if (p.splitk == 1) { p.bias = nullptr; p.act = VT_ACT_NONE; }
  return vt_gemm_launch(p, c.s);
}

int gn(Ctx& c, int nsl, int M_rows_T, int C, const float* cb, const float* g, const float* be, long film_off, bool film,
       const View* res, View out) {
  const vt_unet_desc& d = c.h->d;
  VtGnParams p;
  memset(&p, 0, sizeof(p));
  p.P = (const float*)(c.ws + c.w.slabs); p.nslabs = nsl; p.slab_stride = c.w.slab_stride;
  p.p_gs = (long)c.B * M_rows_T * C; p.ldp = C;
  p.bias = cb; p.gamma = g; p.beta = be; p.vec_gs = C;
  if (film) { p.film = (const float*)(c.ws + c.w.film); p.film_ld = c.h->F; p.film_off = film_off; p.film_gs = (long)c.B * c.h->F; }
  if (res) { p.residual = res->p; p.ldr = res->ld; p.r_gs = res->gs; }
  p.out = out.p; p.ldo = out.ld; p.o_gs = out.gs; p.out_dtype = c.adt;
  p.B = c.B; p.T = M_rows_T; p.C = C; p.ngroups = d.n_groups; p.nets = d.nets; p.eps = 1e-5f;
  return vt_k_groupnorm(p, c.s);
}

#define CK(x) do { int _r = (x); if (_r) return _r; } while (0)

int resblock(Ctx& c, const ResBlk& r, View x, int Tl, View out) {
  const vt_unet_desc& d = c.h->d;
  const long M = (long)c.B * Tl;
  View hbuf = {c.ws + c.w.bufH, r.cout, M * r.cout};
  int nsl;
  CK(conv_slabs(c, x, r.cin_pad, r.c0_w, r.cout, d.ksize, Tl, Tl, 1, -(d.ksize / 2), &nsl));
  CK(gn(c, nsl, Tl, r.cout, r.c0_b, r.g0, r.be0, r.film_off, true, nullptr, hbuf));
  CK(conv_slabs(c, hbuf, r.cout, r.c1_w, r.cout, d.ksize, Tl, Tl, 1, -(d.ksize / 2), &nsl));
  View res = x;
  if (r.res_w) {
    VtGemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = x.p; p.W = r.res_w; p.C = c.ws + c.w.bufR;
    p.M = (int)M; p.N = r.cout; p.K = r.cin_pad; p.lda = x.ld; p.ldw = r.cin_pad; p.ldc = r.cout;
    p.bias = r.res_b; p.bias_gs = r.cout;
    p.groups = d.nets; p.splitk = 1; p.a_gs = x.gs; p.w_gs = (long)r.cout * r.cin_pad; p.c_gs = M * r.cout;
    p.a_dtype = c.adt; p.w_dtype = c.cdt; p.c_dtype = c.adt;
    CK(vt_gemm_launch(p, c.s));
    res = View{c.ws + c.w.bufR, r.cout, M * r.cout};
  }
  CK(gn(c, nsl, Tl, r.cout, r.c1_b, r.g1, r.be1, 0, false, &res, out));
  return VT_OK;
}

// FiLM inputs: g[net][b] = mish(cat(step_mlp(sinusoid(t)), cond));  film = g @ film_w^T + film_b
int film_tables(Ctx& c, const float* t_dev, float t_host, bool cond_ready, const float* cond) {
  const vt_unet_s* h = c.h;
  const vt_unet_desc& d = h->d;
  const int G = d.dsed + d.cond_dim;
  CK(vt_k_sinusoid(t_dev, t_host, c.ws + c.w.sin, c.adt, c.B, d.dsed, 1, 0, 0, c.s));
  VtGemmParams p;
  memset(&p, 0, sizeof(p));
  p.groups = d.nets; p.splitk = 1; p.a_dtype = c.adt; p.w_dtype = c.cdt; p.c_dtype = c.adt;
  // Linear(dsed -> 4 dsed) + Mish
  p.A = c.ws + c.w.sin; p.a_gs = 0; p.lda = d.dsed;
  p.W = h->step_w1; p.w_gs = (long)4 * d.dsed * d.dsed; p.ldw = d.dsed;
  p.bias = h->step_b1; p.bias_gs = 4 * d.dsed; p.act = VT_ACT_MISH;
  p.C = c.ws + c.w.h1; p.c_gs = (long)c.B * 4 * d.dsed; p.ldc = 4 * d.dsed;
  p.M = c.B; p.N = 4 * d.dsed; p.K = d.dsed;
  CK(vt_gemm_launch(p, c.s));
  // Linear(4 dsed -> dsed), then the FiLM cond_encoder's leading Mish, written into g[:, :dsed]
  p.A = c.ws + c.w.h1; p.a_gs = (long)c.B * 4 * d.dsed; p.lda = 4 * d.dsed;
  p.W = h->step_w2; p.w_gs = (long)4 * d.dsed * d.dsed; p.ldw = 4 * d.dsed;
  p.bias = h->step_b2; p.bias_gs = d.dsed; p.act = VT_ACT_MISH;
  p.C = c.ws + c.w.gin; p.c_gs = (long)c.B * G; p.ldc = G;
  p.M = c.B; p.N = d.dsed; p.K = 4 * d.dsed;
  CK(vt_gemm_launch(p, c.s));
  if (!cond_ready) {
    for (int n = 0; n < d.nets; ++n)
      CK(vt_k_act_copy(cond, VT_F32, d.cond_dim, c.ws + c.w.gin + ((size_t)n * c.B * G + d.dsed) * c.a_es, c.adt, G, c.B, d.cond_dim, VT_ACT_MISH, c.s));
  }
  p.A = c.ws + c.w.gin; p.a_gs = (long)c.B * G; p.lda = G;
  p.W = h->film_w; p.w_gs = h->F * G; p.ldw = G;
  p.bias = h->film_b; p.bias_gs = h->F; p.act = VT_ACT_NONE;
  p.C = c.ws + c.w.film; p.c_gs = (long)c.B * h->F; p.ldc = h->F; p.c_dtype = VT_F32;
  p.M = c.B; p.N = (int)h->F; p.K = G;
  CK(vt_gemm_launch(p, c.s));
  return VT_OK;
}

// the convolutional trunk; FiLM tables must be ready; x = fp32 [B][T][input_dim]; out fp32 [nets][B*T][input_dim]
int trunk(Ctx& c, const float* x, float* out) {
  const vt_unet_s* h = c.h;
  const vt_unet_desc& d = h->d;
  const int L = d.n_levels, a = c.a_es;
  const long M0 = (long)c.B * c.T;
  CK(vt_k_pad_cols(x, d.input_dim, c.ws + c.w.xp, c.adt, d.input_pad, M0, c.s));
  View cur = {c.ws + c.w.xp, d.input_pad, 0};
  char* ping = c.ws + c.w.bufX;
  char* pong = c.ws + c.w.bufY;
  int rbi = 0;
  int Tl = c.T;
  for (int l = 0; l < L; ++l) {
    const int C = d.dims[l];
    const long M = (long)c.B * Tl;
    View o1 = {ping, C, M * C};
    CK(resblock(c, h->rb[rbi++], cur, Tl, o1));
    View o2;
    if (l >= 1) {   // skip tensor: second half of the concat buffer of up level u = L-1-l
      const int u = L - 1 - l;
      o2 = View{c.ws + c.w.cat[u] + (size_t)C * a, 2 * C, M * 2 * C};
    } else {
      o2 = View{pong, C, M * C};
    }
    CK(resblock(c, h->rb[rbi++], o1, Tl, o2));
    cur = o2;
    if (l < L - 1) {   // Downsample1d: Conv1d(C, C, 3, stride 2, pad 1)
      VtGemmParams p;
      memset(&p, 0, sizeof(p));
      p.A = cur.p; p.lda = cur.ld; p.a_gs = cur.gs;
      p.W = h->down_w[l]; p.ldw = 3 * C; p.w_gs = (long)C * 3 * C;
      p.bias = h->down_b[l]; p.bias_gs = C;
      p.M = (int)(M / 2); p.N = C; p.K = 3 * C;
      p.taps = 3; p.cin = C; p.tout = Tl / 2; p.tin = Tl; p.stride = 2; p.off0 = -1; p.tstep = 1;
      p.C = ping; p.ldc = C; p.c_gs = (M / 2) * C;
      p.groups = d.nets; p.splitk = 1; p.a_dtype = c.adt; p.w_dtype = c.cdt; p.c_dtype = c.adt;
      CK(vt_gemm_launch(p, c.s));
      // next level reads `ping`; keep ping/pong roles by swapping
      cur = View{ping, C, (M / 2) * C};
      char* t = ping; ping = pong; pong = t;
      Tl /= 2;
    }
  }
  {  // mid blocks; the second writes the first half of cat[0]
    const int C = d.dims[L - 1];
    const long M = (long)c.B * Tl;
    char* free_buf = (cur.p == ping) ? pong : ping;
    if (cur.p != ping && cur.p != pong) free_buf = ping;
    View o1 = {free_buf, C, M * C};
    CK(resblock(c, h->rb[rbi++], cur, Tl, o1));
    View o2 = {c.ws + c.w.cat[0], 2 * C, M * 2 * C};
    CK(resblock(c, h->rb[rbi++], o1, Tl, o2));
  }
  for (int u = 0; u < L - 1; ++u) {
    const int din = d.dims[L - 2 - u], dout = d.dims[L - 1 - u];
    const long M = (long)c.B * Tl;
    View cat = {c.ws + c.w.cat[u], 2 * dout, M * 2 * dout};
    View o1 = {c.ws + c.w.bufX, din, M * din};
    CK(resblock(c, h->rb[rbi++], cat, Tl, o1));
    View o2 = {c.ws + c.w.bufY, din, M * din};
    CK(resblock(c, h->rb[rbi++], o1, Tl, o2));
    // Upsample1d: ConvTranspose1d(din, din, 4, stride 2, pad 1) as two parity GEMMs over 2 taps each
    char* dst; long dld, dgs;
    if (u + 1 < L - 1) { const int cn = d.dims[L - 2 - u]; dst = c.ws + c.w.cat[u + 1]; dld = 2 * cn; dgs = 2 * M * dld; }
    else { dst = c.ws + c.w.bufX; dld = din; dgs = 2 * M * din; }
    for (int par = 0; par < 2; ++par) {
      VtGemmParams p;
      memset(&p, 0, sizeof(p));
      p.A = o2.p; p.lda = o2.ld; p.a_gs = o2.gs;
      p.W = par == 0 ? h->up_we[u] : h->up_wo[u]; p.ldw = 2 * din; p.w_gs = (long)din * 2 * din;
      p.bias = h->up_b[u]; p.bias_gs = din;
      p.M = (int)M; p.N = din; p.K = 2 * din;
      p.taps = 2; p.cin = din; p.tout = Tl; p.tin = Tl; p.stride = 1; p.off0 = par == 0 ? 0 : 1; p.tstep = -1;
      p.C = dst + (size_t)par * dld * a; p.ldc = 2 * dld; p.c_gs = dgs;
      p.groups = d.nets; p.splitk = 1; p.a_dtype = c.adt; p.w_dtype = c.cdt; p.c_dtype = c.adt;
      CK(vt_gemm_launch(p, c.s));
    }
    Tl *= 2;
  }
  {  // final: Conv1dBlock(C0, C0) -> Conv1d(C0, input_dim, 1)
    const int C = d.dims[0];
    const long M = (long)c.B * Tl;
    View xin = {c.ws + c.w.bufX, C, M * C};
    int nsl;
    CK(conv_slabs(c, xin, C, h->fc_w, C, d.ksize, Tl, Tl, 1, -(d.ksize / 2), &nsl));
    View hb = {c.ws + c.w.bufH, C, M * C};
    CK(gn(c, nsl, Tl, C, h->fc_b, h->fg, h->fbe, 0, false, nullptr, hb));
    VtGemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = hb.p; p.lda = C; p.a_gs = M * C;
    p.W = h->out_w; p.ldw = C; p.w_gs = (long)d.input_dim * C;
    p.bias = h->out_b; p.bias_gs = d.input_dim;
    p.M = (int)M; p.N = d.input_dim; p.K = C;
    p.C = out; p.ldc = d.input_dim; p.c_gs = M * d.input_dim;
    p.groups = d.nets; p.splitk = 1; p.a_dtype = c.adt; p.w_dtype = c.cdt; p.c_dtype = VT_F32;
    CK(vt_gemm_launch(p, c.s));
  }
  return VT_OK;
}

int make_ctx(Ctx& c, vt_unet_t h, int B, int T, void* ws, vt_stream_t s) {
  if (!h || !ws) return vt_fail(VT_ERR_ARG, "unet: null handle/workspace");
  if (B <= 0 || T <= 0 || (T % (1 << (h->d.n_levels - 1)))) return vt_fail(VT_ERR_ARG, "unet: T=%d must be divisible by %d", T, 1 << (h->d.n_levels - 1));
  c.h = h; c.B = B; c.T = T; c.ws = (char*)ws; c.w = carve(h, B, T); c.s = (hipStream_t)s;
  c.adt = h->d.adt; c.cdt = h->d.cdt; c.a_es = es(h->d.adt);
  return VT_OK;
}

}  // namespace

// VLATOUCH_UNET_FUSED=0 / vt_tune(7, 0) keeps the sampler on the launch-per-op driver below (A/B)
static int g_unet_fused = -1;
static bool fused_enabled() {
  if (g_unet_fused < 0) { const char* e = getenv("VLATOUCH_UNET_FUSED"); g_unet_fused = (!e || atoi(e) != 0) ? 1 : 0; }
  return g_unet_fused != 0;
}
void vt_unet_fused_tune(int on) { g_unet_fused = on ? 1 : 0; }

size_t vt_unet_fused_plan_bytes(vt_unet_t h, int B, int T, int n_steps) { return h ? vt_unet_fused_workspace_bytes(h, B, T, n_steps) : 0; }
int vt_unet_fused_covers(vt_unet_t h, int B, int T, int n_steps) { return (h && fused_enabled() && vt_unet_fused_ok(h, B, T, n_steps)) ? 1 : 0; }

// the larger of the two drivers' needs; the fused plan is sized whenever the CONFIGURATION supports it, packed or not (a caller may size first and
// pack afterwards)
size_t vt_unet_workspace_bytes(vt_unet_t h, int B, int T) {
  if (!h) return 0;
  const size_t a = carve(h, B, T).total, b = vt_unet_fused_workspace_bytes(h, B, T, 64);
  return a > b ? a : b;
}

int vt_unet_forward(vt_unet_t h, const float* x, const float* t_dev, float t_host, const float* cond, float* out,
                    int B, int T, void* workspace, vt_stream_t stream) {
  Ctx c;
  CK(make_ctx(c, h, B, T, workspace, stream));
  if (!t_dev && fused_enabled() && vt_unet_fused_ok(h, B, T, 1))
    return vt_unet_fused_run(h, const_cast<float*>(x), cond, &t_host, nullptr, 1, nullptr, nullptr, out, B, T, workspace, (hipStream_t)stream);
  CK(vt_wrap(film_tables(c, t_dev, t_host, false, cond), "unet film tables"));
  CK(vt_wrap(trunk(c, x, out), "unet trunk"));
  return VT_OK;
}

// fp32 scalar schedules evaluated the way torch evaluates them on an fp32 tensor (bridge_model.py:59-101)
static int si_schedule(int gamma_type, int epsilon_type, float t, float* gam, float* gder, float* ginv, float* eps) {
  const float GMAX = 200.0f;
  switch (gamma_type) {
    case 0:   // '2^0.5*t(t-1)'
      *gam = 1.4142f * t * (1.0f - t);
      *gder = 1.4142f * (1.0f - 2.0f * t);
      *ginv = 1.0f / (1.4142f * t * (1.0f - t) + 1e-4f);
      break;
    case 1:   // '(2t(t-1))^0.5'
      *gam = 1.4142f * sqrtf(t * (1.0f - t));
      *gder = (1.0f - 2.0f * t) / sqrtf(2.0f * (t - t * t) + 1e-4f);
      *ginv = 1.0f / (1.4142f * sqrtf(t * (1.0f - t) + 1e-4f));
      break;
    case 2:   // '(1-t)^2(2t)^0.5'
      *gam = 1.4142f * ((1.0f - t) * (1.0f - t)) * sqrtf(t);
      *gder = 1.4142f * (2.0f * (t - 1.0f) * sqrtf(t) + ((1.0f - t) * (1.0f - t)) / (2.0f * sqrtf(t + 1e-4f)));
      *ginv = 1.0f / (1.4142f * ((1.0f - t) * (1.0f - t)) * sqrtf(t) + 1e-4f);
      break;
    default: return VT_ERR_UNSUPPORTED;
  }
  *ginv = fminf(fmaxf(*ginv, 0.0f), GMAX);
  switch (epsilon_type) {
    case 0: *eps = (1.0f - t) * 1.0f; break;          // '1-t'
    case 1: *eps = t * (1.0f - t); break;             // 't(t-1)'
    case 2: *eps = 1.0f - sqrtf(t); break;            // '1-sqrt(t)'
    case 3: *eps = 1.0f - t * t; break;               // '1-t^2'
    case 4: *eps = t * 0.0f; break;                   // '0'
    default: return VT_ERR_UNSUPPORTED;
  }
  return VT_OK;
}

int vt_si_sample_ex(vt_unet_t h, float* x, const float* cond, const float* noise, int n_steps, float beta_max, int gamma_type,
                    int epsilon_type, int sde_type, int backward, float score_weight, float* traj, int B, int T, void* workspace, vt_stream_t stream) {
  Ctx c;
  CK(make_ctx(c, h, B, T, workspace, stream));
  if (h->d.nets != 2) return vt_fail(VT_ERR_ARG, "vt_si_sample needs a 2-net handle ({v_net|b_net}, s_net)");
  if (sde_type != 0 && sde_type != 1) return vt_fail(VT_ERR_UNSUPPORTED, "vt_si_sample: sde_type must be 0 ('vs') or 1 ('bs')");
  if (n_steps < 1) return vt_fail(VT_ERR_ARG, "n_steps < 1");
  const long n = (long)B * T * h->d.input_dim;
  hipStream_t s = (hipStream_t)stream;
  float* vs = (float*)(c.ws + c.w.vs_out);
  if (traj) { if (hipMemcpyAsync(traj, x, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "traj copy"); }
  // delta_t = float(1/diffuse_step); n_steps = int(1/delta_t)  (bridge_model.py:335)
  const float dt = (float)(1.0 / (double)n_steps);
  const bool fused = fused_enabled() && vt_unet_fused_ok(h, B, T, n_steps);
  float ts[64];
  VtSdeCoef coef[64];
  for (int k = 1; k <= n_steps; ++k) {
    float t = (float)((double)k / (double)n_steps);
    t = fminf(fmaxf(t, 0.001f), 1.0f - 0.001f);        // t_min clip (bridge_model.py:347-348)
    // direction='backward' (:356-361, :379-382): nets, gamma, gamma', gamma^-1, the noise scale and the score weight at 1 - t; the
    // epsilon inside b stays at t (:369 reads t_tensor in both directions)
    const float tn = backward ? 1.0f - t : t;
    float gam, gder, ginv, eps_n, eps_t, dummy;
    if (si_schedule(gamma_type, epsilon_type, tn, &gam, &gder, &ginv, &eps_n)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_si_sample: unknown gamma/epsilon type");
    eps_t = eps_n;
    if (backward && si_schedule(gamma_type, epsilon_type, t, &dummy, &dummy, &dummy, &eps_t)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_si_sample: unknown gamma/epsilon type");
    const float noise_scale = dt * sqrtf(2.0f * eps_n);
    // 'vs': b = v - gamma_dot*gamma * (s*gamma_inv) * eps (:369);  'bs': b = b_net output (:306) -> no correction term
    const float gdg = sde_type == 0 ? gder * gam : 0.0f;
    if (fused) {
      ts[k - 1] = tn;
      coef[k - 1] = VtSdeCoef{dt, ginv, gdg, eps_t, noise_scale, beta_max, score_weight * eps_n, backward ? 1 : 0};
      continue;
    }
    CK(vt_wrap(film_tables(c, nullptr, tn, k > 1, cond), "si film tables"));
    CK(vt_wrap(trunk(c, x, vs), "si trunk"));
    CK(vt_k_sde_update(x, vs, vs + n, noise ? noise + (long)(k - 1) * n : nullptr, n, dt, ginv, gdg, eps_t, noise_scale, beta_max, score_weight * eps_n,
                       backward ? 1 : 0, s));
    if (traj) { if (hipMemcpyAsync(traj + (long)k * n, x, n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "traj copy"); }
  }
  if (fused) return vt_unet_fused_run(h, x, cond, ts, coef, n_steps, noise, traj, nullptr, B, T, workspace, s);
  return VT_OK;
}

int vt_si_sample(vt_unet_t h, float* x, const float* cond, const float* noise, int n_steps, float beta_max, int gamma_type,
                 int epsilon_type, int sde_type, float* traj, int B, int T, void* workspace, vt_stream_t stream) {
  return vt_si_sample_ex(h, x, cond, noise, n_steps, beta_max, gamma_type, epsilon_type, sde_type, 0, 1.0f, traj, B, T, workspace, stream);
}
