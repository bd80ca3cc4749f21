// vt_lstm.hip — the LSTM residual head as ONE persistent kernel per call: force MLP -> L-layer LSTM cell with carried (h, c) ->
// [h_top | obs_cond] -> Linear -> LayerNorm -> GELU -> Linear -> vla_n + delta, for T consecutive control ticks
// (replaces TactileLSTMController.predict / predict_sequence / forward, residual_controller/lstm_step_controller.py:66-82 (modules),
//  :170-213, :232-286, :288-319).
//
// The head is 1.2 M parameters driven at batch <= a few dozen rows: a dependent chain of tiny matrix products, i.e. latency, not
// FLOPs.  The previous driver issued 17 launches per tick (272 per 16-tick chunk); here a block of 8 waves owns 16 batch rows
// for the WHOLE call:
//   * (h, c): h of every layer lives in LDS (it is the next product's operand), c in registers, across all ticks; global memory
//     sees them once at entry and once at exit;
//   * every Linear is out[16, N] = X[16, K] W[N, K]^T on the MFMA pipe with the 16 rows as the second operand (D[n][m]: a lane ends
//     with 4 consecutive n of one row m); X comes from LDS, W is STREAMED from L2 straight into registers — the weights are
//     pre-packed at load (vlatouch/engine.py) in MFMA fragment order, [n-tile][k-step][lane][8], so a wave's load instruction
//     reads 1 KiB (bf16) / 2 KiB (fp32) contiguous: full-line, coalesced, no LDS staging for an operand no other wave shares;
//   * the gate rows are dealt so that wave w holds all four gates (i, f, g, o) of hidden units [32w, 32w+32) in the SAME lanes:
//     the cell update needs no exchange; only h goes through LDS (one barrier pair per layer);
//   * LayerNorm of the head is a two-row-per-wave reduction over the LDS copy.
// fp32 mode: exact fp32 MFMA (v_mfma_f32_16x16x4_f32) on fp32 weights — 77 MFLOP per tick on the 2 CUs a batch of 32 occupies is then
// the bound (62 us per tick).  x3 mode (the low-precision mode of the head): the same fp32 weights pre-split into bf16 hi + lo
// fragments, 3 bf16 MFMAs per product with fp32 accumulate: fp32-class accuracy (the recurrence keeps its 1e-4 parity) at 5x the
// MFMA rate.  bf16 mode: bf16 weights, the fp32 activations rounded to bf16 at the MFMA input.
#include <math.h>
#include <string.h>
#include <new>
#include "vt_common.h"
#include "vt_host.h"
#include "../../include/vlatouch.h"

namespace {

// LSTM hidden size H (the reference's --hidden_dim, lstm_step_controller.py:16; every width of the head scales with it: force MLP H/2, LSTM H,
// head 2H -> H) is a template parameter: 128, 256 (the reference's default) or 384 — multiples of 128, so that each of the 8 waves owns H/8 =
// UT x 16 hidden units with all four gates in the same lanes.  (512 would need 170 KiB of LDS for two layers.)
constexpr int ROWS = 16;               // batch rows per block
constexpr int KF = 32, LDF = KF + 4;   // raw force, padded to one k-step
template <int H> struct LstmDims {
  static constexpr int UT = H / 128;                                   // 16-unit tiles per gate per wave
  static constexpr int LDH = H + 4;                                    // fp32 LDS row pitches: (pitch mod 64) == 4 keeps the 16-row ds_read_b128 pattern conflict-free
  static constexpr int KX = (H / 2 + 16 + 31) / 32 * 32, LDX = KX + 4; // layer-0 input [force features H/2 | vla_n S <= 16 | 0 ...] padded to whole k-steps (H = 256: 160)
  static constexpr int HF = H / 2, LDF1 = HF + 4;                      // force-MLP width
};

struct LstmSeqParams {
  const void *fe1, *fe2, *wl[4], *h1, *h2;
  const float *fe_b1, *fe_b2, *bl[4], *h1_b, *ln_w, *ln_b, *h2_b;
  const float *obs, *vla, *force;
  float *h, *c, *out;
  int B, T, layers, S, F;
};

struct x3w_t {};                       // fp32 weights pre-split into bf16 hi + lo fragments: a_hi w_hi + a_lo w_hi + a_hi w_lo on the bf16 pipe
template <typename TW> struct WFrag;
template <> struct WFrag<bf16_t> { static constexpr int BYTES = 16; };
template <> struct WFrag<float> { static constexpr int BYTES = 32; };
// x3w_t fragments: [hi 16 B][lo 16 B] per lane (read directly in mm)

// packed weight fragment (tile, ks) of a matrix with `ksteps` k-steps: [tile][ks][lane][8]
template <typename TW>
__device__ __forceinline__ void load_w(Frag<TW>& f, const void* P, int tile, int ksteps, int ks, int lane) {
  const char* p = reinterpret_cast<const char*>(P) + (((long)tile * ksteps + ks) * 64 + lane) * WFrag<TW>::BYTES;
  if constexpr (sizeof(TW) == 2) {
    f.v = *reinterpret_cast<const short8_t*>(p);
  } else {
    const float4_t lo = *reinterpret_cast<const float4_t*>(p), hi = *reinterpret_cast<const float4_t*>(p + 16);
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3]; f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
  }
}
// activation fragment: row l15 of the fp32 LDS matrix X (pitch ld), columns ks*32 + g*8 .. +7
template <typename TW>
__device__ __forceinline__ void load_x(Frag<TW>& f, const float* X, int ld, int ks, int lane) {
  const float* p = X + (lane & 15) * ld + ks * 32 + (lane >> 4) * 8;
  const float4_t lo = *reinterpret_cast<const float4_t*>(p), hi = *reinterpret_cast<const float4_t*>(p + 4);
  if constexpr (sizeof(TW) == 2) {
    uint4 w;
    w.x = pk_bf16(lo[0], lo[1]); w.y = pk_bf16(lo[2], lo[3]); w.z = pk_bf16(hi[0], hi[1]); w.w = pk_bf16(hi[2], hi[3]);
    f.v = __builtin_bit_cast(short8_t, w);
  } else {
    f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3]; f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
  }
}
// acc[t] += W tiles (tile0 + t) x X over k-steps [0, nks) of X, which are the k-steps [wk0, wk0 + nks) of the packed matrix
template <typename TW, int NT>
__device__ __forceinline__ void mm(float4_t (&acc)[NT], const void* P, int tile0, int w_ksteps, int wk0, const float* X, int ld, int nks, int lane) {
#pragma unroll 1
  for (int ks = 0; ks < nks; ++ks) {
    if constexpr (std::is_same<TW, x3w_t>::value) {
      // split-bf16: fp32-class accuracy (the dropped a_lo w_lo term is 2^-16 relative) at 3 bf16 MFMAs instead of 8 fp32 ones
      Frag<bf16_t> whi[NT], wlo[NT], xhi, xlo;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const char* q = reinterpret_cast<const char*>(P) + ((((long)(tile0 + t) * w_ksteps + wk0 + ks) * 64 + lane) * 32);
        whi[t].v = *reinterpret_cast<const short8_t*>(q);
        wlo[t].v = *reinterpret_cast<const short8_t*>(q + 16);
      }
      const float* xp = X + (lane & 15) * ld + ks * 32 + (lane >> 4) * 8;
      const float4_t lo4 = *reinterpret_cast<const float4_t*>(xp), hi4 = *reinterpret_cast<const float4_t*>(xp + 4);
      float xv[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]}, xr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[j] = xv[j] - bf2f(f2bf(xv[j]));
      uint4 a, b;
      a.x = pk_bf16(xv[0], xv[1]); a.y = pk_bf16(xv[2], xv[3]); a.z = pk_bf16(xv[4], xv[5]); a.w = pk_bf16(xv[6], xv[7]);
      b.x = pk_bf16(xr[0], xr[1]); b.y = pk_bf16(xr[2], xr[3]); b.z = pk_bf16(xr[4], xr[5]); b.w = pk_bf16(xr[6], xr[7]);
      xhi.v = __builtin_bit_cast(short8_t, a); xlo.v = __builtin_bit_cast(short8_t, b);
#pragma unroll
      for (int t = 0; t < NT; ++t) { mma16(acc[t], wlo[t], xhi); mma16(acc[t], whi[t], xlo); mma16(acc[t], whi[t], xhi); }
    } else {
      Frag<TW> xf, wf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) load_w<TW>(wf[t], P, tile0 + t, w_ksteps, wk0 + ks, lane);
      load_x<TW>(xf, X, ld, ks, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) mma16(acc[t], wf[t], xf);
    }
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp(2.0f * x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename TW, int NL, int H>
__global__ __launch_bounds__(512) void lstm_seq_kernel(const LstmSeqParams p) {
  using Dm = LstmDims<H>;
  constexpr int UT = Dm::UT, LDH = Dm::LDH, KX = Dm::KX, LDX = Dm::LDX, HF = Dm::HF, LDF1 = Dm::LDF1, UW = H / 8;   // UW = hidden units per wave
  __shared__ __attribute__((aligned(16))) float hS[NL][ROWS][LDH];    // h of every layer
  __shared__ __attribute__((aligned(16))) float obsS[ROWS][LDH];
  __shared__ __attribute__((aligned(16))) float hdS[ROWS][LDH];
  __shared__ __attribute__((aligned(16))) float xinS[ROWS][LDX];
  __shared__ __attribute__((aligned(16))) float f1S[ROWS][LDF1];
  __shared__ __attribute__((aligned(16))) float finS[ROWS][LDF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int b0 = blockIdx.x * ROWS;
  constexpr int L = NL;
  const int S = p.S, F = p.F;
  const int brow = min(b0 + l15, p.B - 1);            // the batch row this lane's MFMA column stands for (clamped: computed, never stored)
  const bool row_ok = b0 + l15 < p.B;

  // ---- entry: h -> LDS, c -> registers (lane owns units j = UW*wave + 16*t + 4*g + r of row l15), obs_cond -> LDS
  float cR[NL][UT][4];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int t = 0; t < UT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) cR[l][t][r] = p.c[((long)l * p.B + brow) * H + wave * UW + t * 16 + g * 4 + r];
  for (int e = tid; e < L * ROWS * (H / 4); e += 512) {
    const int l = e / (ROWS * (H / 4)), rem = e - l * (ROWS * (H / 4)), m = rem / (H / 4), c4 = rem - m * (H / 4);
    *reinterpret_cast<float4*>(&hS[l][m][c4 * 4]) = *reinterpret_cast<const float4*>(p.h + ((long)l * p.B + min(b0 + m, p.B - 1)) * H + c4 * 4);
  }
  for (int e = tid; e < ROWS * (H / 4); e += 512) {
    const int m = e / (H / 4), c4 = e - m * (H / 4);
    *reinterpret_cast<float4*>(&obsS[m][c4 * 4]) = *reinterpret_cast<const float4*>(p.obs + (long)min(b0 + m, p.B - 1) * H + c4 * 4);
  }
  // biases of the gate rows this lane owns: n = q*256 + 32*wave + 16*t + 4*g + r
  for (int e = tid; e < ROWS * LDX; e += 512) (&xinS[0][0])[e] = 0.f;
  for (int e = tid; e < ROWS * LDF; e += 512) (&finS[0][0])[e] = 0.f;
  __syncthreads();

  for (int tk = 0; tk < p.T; ++tk) {
    // ---- inputs of the tick: raw force -> finS, vla_n -> xinS[:, H/2 : H/2 + S]
    if (tid < ROWS * 32) {
      const int m = tid >> 5, k = tid & 31;
      const long bt = (long)min(b0 + m, p.B - 1) * p.T + tk;
      if (k < F) finS[m][k] = p.force[bt * F + k];
      if (k < S) xinS[m][HF + k] = p.vla[bt * S + k];
    }
    __syncthreads();
    // ---- force MLP: Linear(F -> H/2) GELU(erf) Linear(H/2 -> H/2); the H/32 column tiles of 16 are dealt round-robin over the 8 waves
    for (int tile = wave; tile < HF / 16; tile += 8) {
      float4_t a[1] = {(float4_t){0.f, 0.f, 0.f, 0.f}};
      mm<TW, 1>(a, p.fe1, tile, 1, 0, &finS[0][0], LDF, 1, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = tile * 16 + g * 4 + r; f1S[l15][n] = gelu_erf(a[0][r] + p.fe_b1[n]); }
    }
    __syncthreads();
    for (int tile = wave; tile < HF / 16; tile += 8) {
      float4_t a[1] = {(float4_t){0.f, 0.f, 0.f, 0.f}};
      mm<TW, 1>(a, p.fe2, tile, HF / 32, 0, &f1S[0][0], LDF1, HF / 32, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int n = tile * 16 + g * 4 + r; xinS[l15][n] = a[0][r] + p.fe_b2[n]; }
    }
    __syncthreads();
    // ---- LSTM layers: gates = [x | h_l] W_cat^T + (b_ih + b_hh); tiles of wave w: 4 UT w + UT q + t <-> rows q*H + UW w + 16t ..
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const int xks = l == 0 ? KX / 32 : H / 32;                 // k-steps of the layer input
      const int wks = xks + H / 32;
      float4_t acc[4 * UT];
#pragma unroll
      for (int i = 0; i < 4 * UT; ++i) acc[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
      mm<TW, 4 * UT>(acc, p.wl[l], wave * 4 * UT, wks, 0, l == 0 ? &xinS[0][0] : &hS[l - 1][0][0], l == 0 ? LDX : LDH, xks, lane);
      mm<TW, 4 * UT>(acc, p.wl[l], wave * 4 * UT, wks, xks, &hS[l][0][0], LDH, H / 32, lane);
      float hn[UT][4];
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = wave * UW + t * 16 + g * 4 + r;
          const float* bl = p.bl[l];
          const float gi = acc[0 * UT + t][r] + bl[j], gf = acc[1 * UT + t][r] + bl[H + j], gg = acc[2 * UT + t][r] + bl[2 * H + j], go = acc[3 * UT + t][r] + bl[3 * H + j];
          const float cn = sigmoidf_(gf) * cR[l][t][r] + sigmoidf_(gi) * tanhf_(gg);
          cR[l][t][r] = cn;
          hn[t][r] = sigmoidf_(go) * tanhf_(cn);
        }
      __syncthreads();                                           // every wave is done reading the old h_l
#pragma unroll
      for (int t = 0; t < UT; ++t)
        *reinterpret_cast<float4*>(&hS[l][l15][wave * UW + t * 16 + g * 4]) = make_float4(hn[t][0], hn[t][1], hn[t][2], hn[t][3]);
      __syncthreads();
    }
    // ---- head: Linear(2H -> H) on [h_top | obs_cond]; wave w computes columns [UW w, UW w + UW)
    {
      float4_t a[UT];
#pragma unroll
      for (int t = 0; t < UT; ++t) a[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
      mm<TW, UT>(a, p.h1, wave * UT, 2 * H / 32, 0, &hS[L - 1][0][0], LDH, H / 32, lane);
      mm<TW, UT>(a, p.h1, wave * UT, 2 * H / 32, H / 32, &obsS[0][0], LDH, H / 32, lane);
#pragma unroll
      for (int t = 0; t < UT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = wave * UW + t * 16 + g * 4 + r; hdS[l15][n] = a[t][r] + p.h1_b[n]; }
    }
    __syncthreads();
    // LayerNorm(eps 1e-5) + GELU(erf), in place: wave w normalises rows 2w and 2w + 1; a lane holds the float4 chunks lane, lane + 64 (H = 128: lanes
    // 0..31 one chunk; 256: one chunk per lane; 384: lanes 0..31 two)
    constexpr int NCH = (H / 4 + 63) / 64;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      float* row = &hdS[wave * 2 + rr][0];
      float4 v[NCH];
      float sm = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c4 = lane + 64 * k;
        v[k] = c4 < H / 4 ? *reinterpret_cast<float4*>(row + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        sm += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      }
      const float mean = wave_sum(sm) * (1.0f / H);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        if (lane + 64 * k < H / 4) {
          v[k] = make_float4(v[k].x - mean, v[k].y - mean, v[k].z - mean, v[k].w - mean);
          q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
      }
      const float rstd = rsqrtf(wave_sum(q) * (1.0f / H) + 1e-5f);
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int c4 = lane + 64 * k;
        if (c4 < H / 4) {
          const float4 w4 = *reinterpret_cast<const float4*>(p.ln_w + c4 * 4), b4 = *reinterpret_cast<const float4*>(p.ln_b + c4 * 4);
          *reinterpret_cast<float4*>(row + c4 * 4) = make_float4(gelu_erf(v[k].x * rstd * w4.x + b4.x), gelu_erf(v[k].y * rstd * w4.y + b4.y),
                                                                 gelu_erf(v[k].z * rstd * w4.z + b4.z), gelu_erf(v[k].w * rstd * w4.w + b4.w));
        }
      }
    }
    __syncthreads();
    // Linear(H -> S) + vla_n: one tile, wave 0
    if (wave == 0) {
      float4_t a[1] = {(float4_t){0.f, 0.f, 0.f, 0.f}};
      mm<TW, 1>(a, p.h2, 0, H / 32, 0, &hdS[0][0], LDH, H / 32, lane);
      if (row_ok) {
        const long bt = (long)(b0 + l15) * p.T + tk;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int n = g * 4 + r; if (n < S) p.out[bt * S + n] = a[0][r] + p.h2_b[n] + xinS[l15][HF + n]; }
      }
    }
    __syncthreads();                                             // finS / xinS / hdS are rewritten by the next tick
  }
  // ---- exit: (h, c) back to global
  if (row_ok) {
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int t = 0; t < UT; ++t) {
          const long o = ((long)l * p.B + b0 + l15) * H + wave * UW + t * 16 + g * 4;
          *reinterpret_cast<float4*>(p.c + o) = make_float4(cR[l][t][0], cR[l][t][1], cR[l][t][2], cR[l][t][3]);
          *reinterpret_cast<float4*>(p.h + o) = *reinterpret_cast<const float4*>(&hS[l][l15][wave * UW + t * 16 + g * 4]);
        }
  }
}

}  // namespace

// ---- LSTM weight order (vt_lstm_create; packed by vlatouch/engine.py::LstmEngine, fragment order [tile][k-step][lane][8]):
//   0 fe1  (8 tiles x 1 k-step: Linear(F -> 128), K padded to 32)      1 fe_b1 [128]
//   2 fe2  (8 x 4: Linear(128 -> 128))                                  3 fe_b2 [128]
//   per layer l: W_cat = [W_ih | W_hh] (64 tiles x (xks + 8) k-steps, xks = 5 for layer 0 (138 -> 160), else 8; tile 8w + 2q + t
//                = rows q*256 + 32w + 16t ..), b_l = b_ih + b_hh [1024]
//   then h1 (16 x 16: Linear(512 -> 256))  h1_b [256]  ln_w [256]  ln_b [256]  h2 (1 x 8: Linear(256 -> S), rows padded to 16)  h2_b [16]
struct vt_lstm_s {
  vt_lstm_desc d;
  LstmSeqParams w;
};
int vt_lstm_num_weights(const vt_lstm_desc* d) { return 4 + 2 * d->layers + 6; }
int vt_lstm_create(const vt_lstm_desc* desc, const void* const* w, int n, vt_lstm_t* out) {
  if (!desc || !w || !out) return vt_fail(VT_ERR_ARG, "vt_lstm_create: null argument");
  const vt_lstm_desc& d = *desc;
  if (d.hidden != 128 && d.hidden != 256 && d.hidden != 384)
    return vt_fail(VT_ERR_UNSUPPORTED, "vt_lstm_create: hidden must be 128, 256 or 384 (got %d): each of the 8 waves owns hidden/8 = whole 16-unit tiles, and 512 "
                   "would not fit the CU's LDS", d.hidden);
  if (d.layers < 1 || d.layers > 4 || d.state_dim < 1 || d.state_dim > 16 || d.force_dim < 1 || d.force_dim > KF)
    return vt_fail(VT_ERR_ARG, "vt_lstm_create: bad descriptor (layers 1..4, state_dim <= 16, force_dim <= %d)", KF);
  if (d.hidden == 384 && d.layers > 3) return vt_fail(VT_ERR_UNSUPPORTED, "vt_lstm_create: hidden 384 with 4 layers does not fit the CU's LDS");
  if (d.cdt != VT_F32 && d.cdt != VT_BF16 && d.cdt != VT_F32X3) return vt_fail(VT_ERR_ARG, "vt_lstm_create: weights must be fp32, split-bf16 (x3) or bf16");
  if (n != vt_lstm_num_weights(desc)) return vt_fail(VT_ERR_ARG, "vt_lstm_create: expected %d weights, got %d", vt_lstm_num_weights(desc), n);
  for (int k = 0; k < n; ++k) if (!w[k]) return vt_fail(VT_ERR_ARG, "vt_lstm_create: weight %d is null", k);
  vt_lstm_s* h = new (std::nothrow) vt_lstm_s();
  if (!h) return vt_fail(-12, "out of host memory");
  h->d = d;
  memset(&h->w, 0, sizeof(h->w));
  int i = 0;
  h->w.fe1 = w[i++]; h->w.fe_b1 = (const float*)w[i++]; h->w.fe2 = w[i++]; h->w.fe_b2 = (const float*)w[i++];
  for (int l = 0; l < d.layers; ++l) { h->w.wl[l] = w[i++]; h->w.bl[l] = (const float*)w[i++]; }
  h->w.h1 = w[i++]; h->w.h1_b = (const float*)w[i++]; h->w.ln_w = (const float*)w[i++]; h->w.ln_b = (const float*)w[i++];
  h->w.h2 = w[i++]; h->w.h2_b = (const float*)w[i++];
  h->w.layers = d.layers; h->w.S = d.state_dim; h->w.F = d.force_dim;
  *out = h;
  return VT_OK;
}
void vt_lstm_destroy(vt_lstm_t h) { delete h; }
size_t vt_lstm_workspace_bytes(vt_lstm_t h, int B) { (void)B; return h ? 256 : 0; }      // everything lives in LDS / registers

int vt_lstm_sequence(vt_lstm_t hd, const float* obs_cond, const float* vla_n, const float* force, float* h, float* c, float* out_n, int B, int T,
                     vt_stream_t stream) {
  if (!hd || !obs_cond || !vla_n || !force || !h || !c || !out_n) return vt_fail(VT_ERR_ARG, "vt_lstm_sequence: null argument");
  if (B < 1 || T < 1) return vt_fail(VT_ERR_ARG, "vt_lstm_sequence: B, T >= 1");
  LstmSeqParams p = hd->w;
  p.obs = obs_cond; p.vla = vla_n; p.force = force; p.h = h; p.c = c; p.out = out_n; p.B = B; p.T = T;
  const dim3 grid((B + ROWS - 1) / ROWS);
  const int cdt = hd->d.cdt;
#define VT_LSTM_GO2(NL, HH) do { if (cdt == VT_BF16) hipLaunchKernelGGL((lstm_seq_kernel<bf16_t, NL, HH>), grid, dim3(512), 0, (hipStream_t)stream, p); \
                                 else if (cdt == VT_F32X3) hipLaunchKernelGGL((lstm_seq_kernel<x3w_t, NL, HH>), grid, dim3(512), 0, (hipStream_t)stream, p); \
                                 else hipLaunchKernelGGL((lstm_seq_kernel<float, NL, HH>), grid, dim3(512), 0, (hipStream_t)stream, p); } while (0)
#define VT_LSTM_GO(NL) do { if (hd->d.hidden == 128) VT_LSTM_GO2(NL, 128); else if (hd->d.hidden == 256) VT_LSTM_GO2(NL, 256); else VT_LSTM_GO2(NL, 384); } while (0)
  switch (hd->d.layers) {
    case 1: VT_LSTM_GO(1); break;
    case 2: VT_LSTM_GO(2); break;
    case 3: VT_LSTM_GO(3); break;
    default: if (hd->d.hidden == 128) VT_LSTM_GO2(4, 128); else VT_LSTM_GO2(4, 256); break;      // (384 x 4 layers is rejected at create: LDS)
  }
#undef VT_LSTM_GO
#undef VT_LSTM_GO2
  return vt_check_launch();
}

int vt_lstm_step(vt_lstm_t hd, const float* obs_cond, const float* vla_n, const float* force, float* h, float* c, float* out_n, int B,
                 void* workspace, vt_stream_t stream) {
  (void)workspace;
  return vt_lstm_sequence(hd, obs_cond, vla_n, force, h, c, out_n, B, 1, stream);
}
