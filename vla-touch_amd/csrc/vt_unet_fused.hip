// vt_unet_fused.hip — fused driver of the conditional 1-D U-Nets inside the interpolant sampler (split-bf16 mode): the same network as
// vt_unet.hip's launch-per-op driver (bridge/networks/conditional_unet_1D.py:194-247, bridge/bridge_model.py:334-387) in 30 launches per
// SDE step instead of 67 — every launch of the dependent chain costs >= 4.7 us whatever it computes (profiles/r03_seq_pi_*.txt):
//   * each Conv1d is ONE vt_uconv launch that resolves its deferred input (split-K slabs + bias + GroupNorm + Mish + FiLM + residual) in
//     its prologue; the 1x1 residual convolution of a res-block rides along with conv0 as extra n-tiles; torch.cat of the up path is two
//     sources of one reduction; ConvTranspose1d(k4, s2, p1) is one launch over both output parities;
//   * the FiLM table of ALL steps is computed once per sample() call: the step embedding depends on the scalar t_k only, the condition
//     part on the sample only, and cond_encoder's Linear is linear in the concatenation: film[k][b] = W_s mish(e_k) + W_c mish(cond_b) + bias;
//   * final_conv.1 (1x1 to the action dims) of both nets and the Euler-Maruyama update are one small kernel (vt_ufinal).
// Res-block outputs that a later launch needs as a plain tensor (identity residuals, skip connections) are written by the first consumer.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_host.h"
#include "vt_uconv.h"
#include "vt_unet_int.h"

namespace {

enum { SA = 0, SB, SR0, SR1, SD, X0, X1, SK0, SK1, SK2, SK3, NBUF };

struct Tile { int J, cs, S, nsamp, mtiles; size_t lds; int l_r, l_st, l_par, l_hi, l_lo, pitch; };

// Tile choice by a small cost model (us per block, measured orders of magnitude): a k-step (32 channels of one tap, 2 KiB of weights per
// wave) costs ~0.1 us — the CU's operand ingest, not the MFMAs; every round of the CONSUMER's prologue gather (16 loads in flight) ~1.5 us, and
// the number of rounds grows with the slices this launch writes; more than two blocks per CU run in waves.
// cpg_min: the smallest GroupNorm group width among the sources (0 = no GroupNorm): bounds the statistics units of a block.
bool pick_tile(int B, int Tin, int Tq, int Ntiles, int ntaps, int Ctot, int cmin, int unit, int cpg_min, int groups, bool has_res_in, Tile* out) {
  static const double t_step = [] { const char* e = getenv("VLATOUCH_UC_TSTEP"); return e ? atof(e) : 0.1; }();
  static const double t_round = [] { const char* e = getenv("VLATOUCH_UC_TROUND"); return e ? atof(e) : 1.5; }();
  static const double blk_cap = [] { const char* e = getenv("VLATOUCH_UC_BLOCKS"); return e ? atof(e) : 512.0; }();
  double best_cost = 1e30;
  Tile bt;
  bool found = false;
  for (int J = 4; J >= 1; --J) {
    if (J == 3 && Tq % 3) continue;                                // 48-row blocks only for levels that are multiples of 3 ticks (T = 48 / 24 / 12 ...)
    if ((16 * J) % Tq && (Tq % 3 || 16 * J < Tq)) continue;        // whole samples per block; levels of 3 x 2^k ticks may pad the block's last rows
    const int nsamp = 16 * J / Tq;
    if (J > 1 && !(J & 1) && nsamp > B && (8 * J) % Tq == 0) continue;        // a smaller tile still holds whole samples: do not pad rows
    const int rows_in = nsamp * Tin;
    const int mtiles = (B + nsamp - 1) / nsamp;
    for (int cs = unit; cs <= cmin && cs <= 256 && rows_in * cs <= 4096; cs *= 2) {
      if (cmin % cs || Ctot % cs) break;
      const int S = Ctot / cs;
      if (S > 64) continue;
      if (cpg_min > 0 && nsamp * (cs / cpg_min) > 256) continue;    // GroupNorm statistics units of a block: the LDS region holds 256
      const long blocks = (long)groups * mtiles * Ntiles * S;
      const int ne = (rows_in * cs + 1023) / 1024;
      const int rounds = ne > 2 ? ((ne + 3) / 4) * ((S + 3) / 4) : ne * ((S + 15) / 16);
      double cost = ntaps * (cs / 32) * (J == 4 ? 1.3 : 1.0) * t_step + t_round * rounds;
      if (blocks > blk_cap) cost *= (double)blocks / blk_cap;
      if (cost < best_cost) {
        best_cost = cost;
        Tile t;
        t.J = J; t.cs = cs; t.S = S; t.nsamp = nsamp; t.mtiles = mtiles;
        t.pitch = cs * 2 + 16;
        size_t o = (size_t)rows_in * cs * 4;
        t.l_r = (int)o; if (has_res_in) o += (size_t)rows_in * cs * 4;
        t.l_st = (int)o; o += 256 * 8;
        t.l_par = (int)o; o += (size_t)(2 + 2 * nsamp) * cs * 4;
        const size_t plane = (size_t)nsamp * (Tin + 4) * t.pitch;
        t.l_hi = (int)o; o += plane;
        t.l_lo = (int)o; o += plane;
        t.lds = o;
        bt = t;
        found = true;
      }
    }
  }
  if (found) *out = bt;
  return found;
}

struct Run {
  const vt_unet_s* h; int B, T, n_steps; char* ws; hipStream_t s; bool dry;
  size_t need[NBUF]; size_t off[NBUF];
  size_t o_film_s, o_film_c, o_sin, o_h1, o_gs, o_gc, total;
  const float* film_s_k;          // FiLM step rows of the current step
  float* buf(int id) const { return reinterpret_cast<float*>(ws + off[id]); }
  void want(int id, size_t bytes) { if (bytes > need[id]) need[id] = bytes; }
};

#define CK(x) do { int _r = (x); if (_r) return _r; } while (0)

USrc src_zero() { USrc u; memset(&u, 0, sizeof(u)); return u; }

// slabs [net][S][rows][C] of a convolution + its bias (+ GroupNorm / FiLM set by the caller)
USrc src_slabs(const Run& R, int bufid, int S, long rows, int C, const float* bias) {
  USrc u = src_zero();
  u.p = R.dry ? nullptr : R.buf(bufid); u.ld = C; u.nslabs = S; u.slab = rows * C; u.gs = (long)S * rows * C;
  u.bias = bias; u.vec_gs = C; u.C = C; u.cvalid = C;
  return u;
}
USrc src_tensor(const Run& R, int bufid, long rows, int C) {
  USrc u = src_zero();
  u.p = R.dry ? nullptr : R.buf(bufid); u.ld = C; u.gs = rows * C; u.C = C; u.cvalid = C; u.vec_gs = C;
  return u;
}

// one fused convolution; *S_out = number of slabs it wrote
int fconv(Run& R, const FConv& fc, USrc s0, const USrc* s1, int mat_buf, int Tin, int Tq, int stride, int npar, int omul, const int* offs /*[npar][6]*/,
          int outbuf, int routbuf, int* S_out) {
  const vt_unet_desc& d = R.h->d;
  const int Ctot = s0.C + (s1 ? s1->C : 0);
  if (Ctot != fc.nc32 * 32) return vt_fail(VT_ERR_ARG, "fused conv: reduction width %d != packed %d", Ctot, fc.nc32 * 32);
  int unit = 32;
  if (s0.cpg > unit) unit = s0.cpg;
  if (s1 && s1->cpg > unit) unit = s1->cpg;
  int cmin = s0.C;
  if (s1 && s1->C < cmin) cmin = s1->C;
  const bool has_res_in = s0.res_mode || (s1 && s1->res_mode);
  Tile t;
  int cpg_min = s0.cpg;
  if (s1 && s1->cpg > 0 && (cpg_min == 0 || s1->cpg < cpg_min)) cpg_min = s1->cpg;
  if (!pick_tile(R.B, Tin, Tq, (fc.N / 64) * (1 + fc.has_res), fc.ntaps, Ctot, cmin, unit, cpg_min, d.nets * npar, has_res_in, &t))
    return vt_fail(VT_ERR_UNSUPPORTED, "fused conv: no tile for T=%d/%d N=%d C=%d", Tin, Tq, fc.N, Ctot);
  const long Mout = (long)R.B * Tq * omul;
  *S_out = t.S;
  R.want(outbuf, (size_t)d.nets * t.S * Mout * fc.N * 4);
  if (routbuf >= 0) R.want(routbuf, (size_t)d.nets * t.S * Mout * fc.N * 4);
  if (mat_buf >= 0) R.want(mat_buf, (size_t)d.nets * R.B * Tin * s0.C * 4);
  if (R.dry) return VT_OK;
  UConvParams p;
  memset(&p, 0, sizeof(p));
  p.src[0] = s0;
  if (mat_buf >= 0) { p.src[0].mat = R.buf(mat_buf); p.src[0].mat_ld = s0.C; p.src[0].mat_gs = (long)R.B * Tin * s0.C; }
  if (s1) p.src[1] = *s1;
  p.c_split = s1 ? s0.C : Ctot;
  p.Wp = fc.wp; p.w_gs = fc.w_gs; p.w_ps = fc.w_ps; p.nc32 = fc.nc32; p.Wr = fc.wr; p.wr_gs = fc.wr_gs;
  p.out = R.buf(outbuf); p.out_slab = Mout * fc.N; p.out_gs = (long)t.S * Mout * fc.N; p.ldc = fc.N;
  p.has_res = fc.has_res;
  if (fc.has_res) {
    if (routbuf < 0) return vt_fail(VT_ERR_ARG, "fused conv: residual weights without an output buffer");
    p.rout = R.buf(routbuf); p.rout_slab = p.out_slab; p.rout_gs = p.out_gs;
  }
  p.nets = d.nets; p.npar = npar; p.B = R.B; p.Tin = Tin; p.Tq = Tq; p.stride = stride; p.ntaps = fc.ntaps; p.omul = omul;
  for (int i = 0; i < npar * 6; ++i) p.off[i] = offs[i];
  p.N = fc.N; p.cs = t.cs; p.S = t.S; p.nsamp = t.nsamp; p.mtiles = t.mtiles; p.ntiles = fc.N / 64;
  p.nw = d.nets * npar * p.ntiles * (1 + fc.has_res) * t.S;
  p.lds_rstage = t.l_r; p.lds_stats = t.l_st; p.lds_par = t.l_par; p.lds_hi = t.l_hi; p.lds_lo = t.l_lo; p.pitch = t.pitch;
  p.eps = 1e-5f;
  return vt_uconv_launch(p, t.J, t.lds, R.s);
}

void set_gn(USrc& u, const float* g, const float* b, int cpg) { u.gamma = g; u.beta = b; u.cpg = cpg; }

// the convolutional trunk on the state x; ends with the final kernel (1x1 conv of both nets + optional SDE update)
int trunk(Run& R, const float* x, const UFinalParams* fin_proto) {
  const vt_unet_s* h = R.h;
  const vt_unet_desc& d = h->d;
  const int L = d.n_levels, pad = d.ksize / 2;
  int offs_k[12] = {0};
  for (int i = 0; i < d.ksize; ++i) offs_k[i] = i - pad;              // taps of Conv1d(k, padding k/2); slot ntaps (the 1x1 residual conv) = 0
  offs_k[d.ksize] = 0;
  const int offs_dn[12] = {-1, 0, 1, 0, 0, 0};                          // Conv1d(3, stride 2, padding 1): rows 2t-1 .. 2t+1
  const int offs_up[12] = {0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0};        // ConvTranspose1d(4, 2, 1): even rows taps (1, 3) <- rows t, t-1; odd rows taps (0, 2) <- t+1, t
  int Tl = R.T;
  int xsel = 0, rsel = 0, rbi = 0;

  USrc cur = src_zero();
  cur.p = x; cur.ld = d.input_dim; cur.gs = 0; cur.cvalid = d.input_dim; cur.C = h->f_c0[0].nc32 * 32; cur.vec_gs = 0;

  // res-block: `in` (optionally concatenated with the plain tensor in2) -> deferred output
  auto resblock = [&](const ResBlk& r, int idx, USrc in, const USrc* in2, int mat_in, USrc* out) -> int {
    const long M = (long)R.B * Tl;
    const bool ident = !r.res_w;
    if (ident && in2) return vt_fail(VT_ERR_UNSUPPORTED, "fused U-Net: concatenated input with an identity residual");
    int matb = mat_in;
    if (ident && in.nslabs > 0 && matb < 0) { matb = X0 + xsel; xsel ^= 1; }
    int S0 = 0, S1 = 0;
    const int rbuf = ident ? -1 : SR0 + rsel;
    CK(fconv(R, h->f_c0[idx], in, in2, matb, Tl, Tl, 1, 1, 1, offs_k, SA, rbuf, &S0));
    USrc hd = src_slabs(R, SA, S0, M, r.cout, r.c0_b);
    set_gn(hd, r.g0, r.be0, r.cout / d.n_groups);
    hd.film_s = R.film_s_k; hd.film_s_gs = (long)R.n_steps * h->F;
    hd.film_c = R.dry ? nullptr : reinterpret_cast<const float*>(R.ws + R.o_film_c); hd.film_c_gs = (long)R.B * h->F; hd.film_ld = h->F;
    hd.film_off = r.film_off; hd.film_C = r.cout;
    if (R.dry) hd.film_s = nullptr;
    CK(fconv(R, h->f_c1[idx], hd, nullptr, -1, Tl, Tl, 1, 1, 1, offs_k, SB, -1, &S1));
    USrc o = src_slabs(R, SB, S1, M, r.cout, r.c1_b);
    set_gn(o, r.g1, r.be1, r.cout / d.n_groups);
    if (ident) {
      o.res_mode = 1;
      if (in.nslabs == 0) { o.res = in.p; o.res_ld = in.ld; o.res_gs = in.gs; }
      else { o.res = R.dry ? nullptr : R.buf(matb); o.res_ld = r.cout; o.res_gs = M * r.cout; }
    } else {
      o.res_mode = 2; o.res = R.dry ? nullptr : R.buf(rbuf); o.res_ld = r.cout; o.res_nslabs = S0; o.res_slab = M * r.cout; o.res_gs = (long)S0 * M * r.cout;
      o.res_bias = r.res_b;
      rsel ^= 1;
    }
    *out = o;
    return VT_OK;
  };

  for (int l = 0; l < L; ++l) {
    const int C = d.dims[l];
    USrc o1, o2;
    CK(resblock(h->rb[rbi], rbi, cur, nullptr, -1, &o1)); ++rbi;
    CK(resblock(h->rb[rbi], rbi, o1, nullptr, -1, &o2)); ++rbi;
    cur = o2;
    if (l < L - 1) {       // Downsample1d; its prologue also writes the skip tensor of this level (levels >= 1 are consumed by the up path)
      int Sd = 0;
      CK(fconv(R, h->f_down[l], cur, nullptr, l >= 1 ? SK0 + l : -1, Tl, Tl / 2, 2, 1, 1, offs_dn, SD, -1, &Sd));
      Tl /= 2;
      cur = src_slabs(R, SD, Sd, (long)R.B * Tl, C, h->down_b[l]);
    }
  }
  {   // mid blocks: the first one's conv0 writes the deepest level's skip tensor (also its own identity residual)
    USrc o1, o2;
    CK(resblock(h->rb[rbi], rbi, cur, nullptr, SK0 + (L - 1), &o1)); ++rbi;
    CK(resblock(h->rb[rbi], rbi, o1, nullptr, -1, &o2)); ++rbi;
    cur = o2;
  }
  for (int u = 0; u < L - 1; ++u) {
    const int din = d.dims[L - 2 - u], dout = d.dims[L - 1 - u];
    const long M = (long)R.B * Tl;
    USrc skip = src_tensor(R, SK0 + (L - 1 - u), M, dout);
    USrc o1, o2;
    CK(resblock(h->rb[rbi], rbi, cur, &skip, -1, &o1)); ++rbi;
    CK(resblock(h->rb[rbi], rbi, o1, nullptr, -1, &o2)); ++rbi;
    int Su = 0;
    CK(fconv(R, h->f_up[u], o2, nullptr, -1, Tl, Tl, 1, 2, 2, offs_up, SD, -1, &Su));
    Tl *= 2;
    cur = src_slabs(R, SD, Su, (long)R.B * Tl, din, h->up_b[u]);
  }
  {   // final_conv.0's convolution, then the final kernel
    const int C = d.dims[0];
    int Sf = 0;
    CK(fconv(R, h->f_fc, cur, nullptr, -1, Tl, Tl, 1, 1, 1, offs_k, SA, -1, &Sf));
    if (R.dry) return VT_OK;
    UFinalParams f = *fin_proto;
    f.slabs = R.buf(SA); f.nslabs = Sf; f.slab = (long)R.B * Tl * C; f.gs = (long)Sf * R.B * Tl * C;
    f.bias = h->fc_b; f.gamma = h->fg; f.beta = h->fbe; f.vec_gs = C; f.cpg = C / d.n_groups;
    f.out_w = reinterpret_cast<const float*>(h->out_w); f.out_b = h->out_b; f.ow_gs = (long)d.input_dim * C; f.ob_gs = d.input_dim;
    f.B = R.B; f.T = Tl; f.C = C; f.dim = d.input_dim; f.nets = d.nets; f.gn_eps = 1e-5f;
    CK(vt_ufinal_launch(f, R.s));
  }
  return VT_OK;
}

int plan(Run& R) {   // dry run: buffer sizes -> offsets
  memset(R.need, 0, sizeof(R.need));
  R.dry = true;
  R.film_s_k = nullptr;
  CK(trunk(R, nullptr, nullptr));
  const vt_unet_s* h = R.h;
  const vt_unet_desc& d = h->d;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
  R.o_film_s = take((size_t)d.nets * R.n_steps * h->F * 4);
  R.o_film_c = take((size_t)d.nets * R.B * h->F * 4);
  R.o_sin = take((size_t)R.n_steps * d.dsed * 4);
  R.o_h1 = take((size_t)d.nets * R.n_steps * 4 * d.dsed * 4);
  R.o_gs = take((size_t)d.nets * R.n_steps * d.dsed * 4);
  R.o_gc = take((size_t)R.B * d.cond_dim * 4);
  for (int i = 0; i < NBUF; ++i) R.off[i] = take(R.need[i]);
  R.total = o;
  R.dry = false;
  return VT_OK;
}

// FiLM tables of all steps: film_s[net][k][F] = film_w[:, :dsed] mish(step_mlp(sinusoid(t_k))) + film_b; film_c[net][b][F] = film_w[:, dsed:] mish(cond_b)
int film_tables(Run& R, const float* ts, const float* cond) {
  const vt_unet_s* h = R.h;
  const vt_unet_desc& d = h->d;
  const int G = d.dsed + d.cond_dim, n = R.n_steps;
  CK(vt_usin_launch(ts, n, reinterpret_cast<float*>(R.ws + R.o_sin), d.dsed, R.s));
  VtGemmParams p;
  memset(&p, 0, sizeof(p));
  p.groups = d.nets; p.splitk = 1; p.a_dtype = VT_F32; p.w_dtype = d.cdt; p.c_dtype = VT_F32;
  p.A = R.ws + R.o_sin; p.a_gs = 0; p.lda = d.dsed;
  p.W = h->step_w1; p.w_gs = (long)4 * d.dsed * d.dsed; p.ldw = d.dsed;
  p.bias = h->step_b1; p.bias_gs = 4 * d.dsed; p.act = VT_ACT_MISH;
  p.C = R.ws + R.o_h1; p.c_gs = (long)n * 4 * d.dsed; p.ldc = 4 * d.dsed;
  p.M = n; p.N = 4 * d.dsed; p.K = d.dsed;
  CK(vt_gemm_launch(p, R.s));
  p.A = R.ws + R.o_h1; p.a_gs = (long)n * 4 * d.dsed; p.lda = 4 * d.dsed;
  p.W = h->step_w2; p.w_gs = (long)4 * d.dsed * d.dsed; p.ldw = 4 * d.dsed;
  p.bias = h->step_b2; p.bias_gs = d.dsed; p.act = VT_ACT_MISH;        // cond_encoder's leading Mish
  p.C = R.ws + R.o_gs; p.c_gs = (long)n * d.dsed; p.ldc = d.dsed;
  p.M = n; p.N = d.dsed; p.K = 4 * d.dsed;
  CK(vt_gemm_launch(p, R.s));
  p.A = R.ws + R.o_gs; p.a_gs = (long)n * d.dsed; p.lda = d.dsed;
  p.W = h->film_w; p.w_gs = h->F * G; p.ldw = G;
  p.bias = h->film_b; p.bias_gs = h->F; p.act = VT_ACT_NONE;
  p.C = R.ws + R.o_film_s; p.c_gs = (long)n * h->F; p.ldc = h->F;
  p.M = n; p.N = (int)h->F; p.K = d.dsed;
  CK(vt_gemm_launch(p, R.s));
  CK(vt_k_act_copy(cond, VT_F32, d.cond_dim, R.ws + R.o_gc, VT_F32, d.cond_dim, R.B, d.cond_dim, VT_ACT_MISH, R.s));
  p.A = R.ws + R.o_gc; p.a_gs = 0; p.lda = d.cond_dim;
  p.W = reinterpret_cast<const float*>(h->film_w) + d.dsed; p.w_gs = h->F * G; p.ldw = G;
  p.bias = nullptr; p.bias_gs = 0;
  p.C = R.ws + R.o_film_c; p.c_gs = (long)R.B * h->F; p.ldc = h->F;
  p.M = R.B; p.N = (int)h->F; p.K = d.cond_dim;
  CK(vt_gemm_launch(p, R.s));
  return VT_OK;
}

bool config_ok(const vt_unet_s* h) {
  const vt_unet_desc& d = h->d;
  if (d.cdt != VT_F32X3 || d.adt != VT_F32) return false;
  if (d.ksize > 5 || !(d.ksize & 1) || d.input_pad > 32 || d.input_dim > 16) return false;
  for (int l = 0; l < d.n_levels; ++l) {
    const int C = d.dims[l], cpg = C / d.n_groups;
    if (C % 64 || C % d.n_groups || (cpg & (cpg - 1)) || cpg > 128 || cpg < 4) return false;
  }
  return true;
}

// Can EVERY launch of the fused plan run for this (B, T)?  T must halve cleanly down the levels (any T <= 64 whose deepest level is >= 1 tick:
// 16, 32, 48, 64, 24, ...), every convolution must find a tile (dry plan: the same pick_tile calls as the real run, no launches), and the final
// kernel's LDS (activations + 1x1 weights of both nets) must fit the CU.  A configuration that fails stays on the launch-per-op driver.
bool shape_ok(const vt_unet_s* h, int B, int T, int n_steps, size_t* ws_bytes) {
  if (B < 1 || n_steps < 1 || n_steps > 64 || T < 1 || T > 64) return false;
  const int L = h->d.n_levels;
  if (T % (1 << (L - 1))) return false;
  if (vt_ufinal_lds_bytes(T, h->d.input_dim, h->d.dims[0]) > 160 * 1024) return false;
  Run R;
  memset(&R, 0, sizeof(R));
  R.h = h; R.B = B; R.T = T; R.n_steps = n_steps;
  if (plan(R) != VT_OK) return false;            // (a failing dry plan only leaves its reason in the thread's last-error string)
  if (ws_bytes) *ws_bytes = R.total;
  return true;
}

}  // namespace

// `h->fused` (the packed weights exist) is required to RUN the fused path, not to size its workspace: a caller that sizes the workspace before it
// packs must get the fused plan's size too (vt_unet_workspace_bytes).
bool vt_unet_fused_ok(const vt_unet_s* h, int B, int T, int n_steps) { return h && h->fused && config_ok(h) && shape_ok(h, B, T, n_steps, nullptr); }

size_t vt_unet_fused_workspace_bytes(const vt_unet_s* h, int B, int T, int n_steps) {
  size_t total = 0;
  if (!h || !config_ok(h) || !shape_ok(h, B, T, n_steps, &total)) return 0;
  return total;
}

int vt_unet_fused_run(const vt_unet_s* h, float* x, const float* cond, const float* ts, const VtSdeCoef* coef, int n_steps, const float* noise,
                      float* traj, float* vs_out, int B, int T, void* ws, hipStream_t s) {
  if (!vt_unet_fused_ok(h, B, T, n_steps)) return vt_fail(VT_ERR_UNSUPPORTED, "fused U-Net path not available for this configuration");
  if (!coef && n_steps != 1) return vt_fail(VT_ERR_ARG, "fused U-Net forward: one step");
  if (coef && h->d.nets != 2) return vt_fail(VT_ERR_ARG, "fused sampler needs two nets");
  Run R;
  memset(&R, 0, sizeof(R));
  R.h = h; R.B = B; R.T = T; R.n_steps = n_steps; R.ws = (char*)ws; R.s = s;
  CK(plan(R));
  CK(vt_wrap(film_tables(R, ts, cond), "fused film tables"));
  const long n = (long)B * T * h->d.input_dim;
  for (int k = 0; k < n_steps; ++k) {
    R.film_s_k = reinterpret_cast<const float*>(R.ws + R.o_film_s) + (long)k * h->F;
    UFinalParams f;
    memset(&f, 0, sizeof(f));
    f.x = x; f.vs = vs_out;
    if (coef) {
      const VtSdeCoef& c = coef[k];
      f.do_sde = 1; f.z = noise ? noise + (long)k * n : nullptr; f.traj = traj ? traj + (long)(k + 1) * n : nullptr;
      f.dt = c.dt; f.gi = c.gi; f.gdg = c.gdg; f.eps_t = c.eps_t; f.noise_scale = c.noise_scale; f.d = c.d; f.score_eps = c.score_eps; f.backward = c.backward;
    }
    CK(vt_wrap(trunk(R, x, &f), "fused trunk"));
  }
  return VT_OK;
}

// ---------------------------------------------------------------------------------------------------------------- C ABI: weight packing
static void conv_pack_dims(const vt_unet_s* h, int which, int idx, int* N, int* ntaps, int* cinp, int* has_res) {
  const vt_unet_desc& d = h->d;
  const int L = d.n_levels;
  *has_res = 0;
  switch (which) {
    case 0: { const ResBlk& r = h->rb[idx]; *N = r.cout; *ntaps = d.ksize; *cinp = r.cin_pad; *has_res = r.res_w ? 1 : 0; break; }
    case 1: { const ResBlk& r = h->rb[idx]; *N = r.cout; *ntaps = d.ksize; *cinp = r.cout; break; }
    case 2: *N = d.dims[idx]; *ntaps = 3; *cinp = d.dims[idx]; break;
    case 3: *N = d.dims[L - 2 - idx]; *ntaps = 2; *cinp = d.dims[L - 2 - idx]; break;
    default: *N = d.dims[0]; *ntaps = d.ksize; *cinp = d.dims[0]; break;
  }
}

static size_t conv_pack_elems(const vt_unet_s* h, int which, int idx) {   // bf16 elements per net (both parities for the up-sample; + the 1x1 stream)
  int N, ntaps, cinp, has_res;
  conv_pack_dims(h, which, idx, &N, &ntaps, &cinp, &has_res);
  const int nc32 = (cinp + 31) / 32;
  return (size_t)(N / 16) * nc32 * (ntaps * (which == 3 ? 2 : 1) + has_res) * 1024;
}

// shapes of every fused convolution (N, taps, packed reduction width, residual stream), known from the descriptor alone: set at vt_unet_create so
// that the plan can be sized (vt_unet_fused_plan_bytes, vt_unet_workspace_bytes) before — or without — vt_unet_fused_pack
void vt_unet_fused_init_meta(vt_unet_s* h) {
  if (!h || !config_ok(h)) return;
  const int L = h->d.n_levels;
  auto meta = [&](FConv& fc, int which, int idx) {
    int N, ntaps, cinp, has_res;
    conv_pack_dims(h, which, idx, &N, &ntaps, &cinp, &has_res);
    fc.nc32 = (cinp + 31) / 32; fc.ntaps = ntaps; fc.has_res = has_res; fc.N = N;
  };
  for (int i = 0; i < h->nrb; ++i) { meta(h->f_c0[i], 0, i); meta(h->f_c1[i], 1, i); }
  for (int l = 0; l < L - 1; ++l) { meta(h->f_down[l], 2, l); meta(h->f_up[l], 3, l); }
  meta(h->f_fc, 4, 0);
}

size_t vt_unet_fused_bytes(vt_unet_t h) {
  if (!h || !config_ok(h)) return 0;
  const int L = h->d.n_levels;
  size_t e = 0;
  for (int i = 0; i < h->nrb; ++i) e += conv_pack_elems(h, 0, i) + conv_pack_elems(h, 1, i);
  for (int l = 0; l < L - 1; ++l) e += conv_pack_elems(h, 2, l) + conv_pack_elems(h, 3, l);
  e += conv_pack_elems(h, 4, 0);
  return e * h->d.nets * 2;
}

int vt_unet_fused_pack(vt_unet_t h, void* buf, vt_stream_t stream) {
  if (!h || !buf) return vt_fail(VT_ERR_ARG, "vt_unet_fused_pack: null argument");
  if (!config_ok(h)) return vt_fail(VT_ERR_UNSUPPORTED, "vt_unet_fused_pack: configuration not supported by the fused path (split-bf16 mode, dims %% 64, power-of-two groups)");
  hipStream_t s = (hipStream_t)stream;
  const vt_unet_desc& d = h->d;
  const int L = d.n_levels, nets = d.nets;
  uint16_t* o = reinterpret_cast<uint16_t*>(buf);
  auto pack = [&](FConv& fc, int which, int idx, const void* Wm, const void* Wr, const void* Wm2) -> int {
    int N, ntaps, cinp, has_res;
    conv_pack_dims(h, which, idx, &N, &ntaps, &cinp, &has_res);
    const int nc32 = (cinp + 31) / 32;
    const size_t per_net = conv_pack_elems(h, which, idx);
    const size_t one = (size_t)(N / 16) * nc32 * ntaps * 1024;            // one parity of the main stream
    fc.wp = o; fc.w_gs = (long)per_net; fc.w_ps = (long)one; fc.nc32 = nc32; fc.ntaps = ntaps; fc.has_res = has_res; fc.N = N;
    fc.wr = nullptr; fc.wr_gs = (long)per_net;
    CK(vt_uconv_pack((const float*)Wm, o, nets, N, ntaps, cinp, nc32, (long)per_net, s));
    if (which == 3) CK(vt_uconv_pack((const float*)Wm2, o + one, nets, N, ntaps, cinp, nc32, (long)per_net, s));
    if (has_res) { fc.wr = o + one; CK(vt_uconv_pack((const float*)Wr, o + one, nets, N, 1, cinp, nc32, (long)per_net, s)); }
    o += per_net * nets;
    return VT_OK;
  };
  for (int i = 0; i < h->nrb; ++i) {
    CK(pack(h->f_c0[i], 0, i, h->rb[i].c0_w, h->rb[i].res_w, nullptr));
    CK(pack(h->f_c1[i], 1, i, h->rb[i].c1_w, nullptr, nullptr));
  }
  for (int l = 0; l < L - 1; ++l) {
    CK(pack(h->f_down[l], 2, l, h->down_w[l], nullptr, nullptr));
    CK(pack(h->f_up[l], 3, l, h->up_we[l], nullptr, h->up_wo[l]));
  }
  CK(pack(h->f_fc, 4, 0, h->fc_w, nullptr, nullptr));
  h->fused = true;
  return VT_OK;
}
