// vt_uconv.hip — fused convolution of the interpolant controller's conditional 1-D U-Nets (replaces, per Conv1d of
// bridge/networks/conditional_unet_1D.py:40-105,194-247, the launch pair "implicit-GEMM conv -> GroupNorm+Mish(+FiLM)(+residual)").
//
//   prologue   resolve the DEFERRED input tensor (vt_uconv.h) for this block's samples and channel slice: sum the producer's split-K slabs
//              + bias, GroupNorm statistics over (T rows x C/8 channels) of each sample, Mish, FiLM, residual; split into bf16 hi / lo and
//              lay the rows out in LDS with a two-row zero halo per sample, so a tap is a row offset of the fragment read
//   k-loop     a wave owns 16 output channels: its weights (frozen, hi / lo pre-split, MFMA fragment order) stream global -> VGPR, one
//              contiguous KiB per load instruction, 8 k-steps in flight; activations are read from LDS only; 3 bf16 MFMAs per 16x16x32
//              product (a_hi w_hi + a_lo w_hi + a_hi w_lo = the split-bf16 arithmetic of vt_gemm.hip); no barrier inside the loop
//   epilogue   raw fp32 slab of this channel slice (the consumer sums the slices)
// The 1x1 residual convolution of a res-block rides along with conv0 as extra n-tiles over the same resolved input (own weight stream, own slabs).
// One block = 16*J output rows (whole samples) x 64 output channels x one slice of `cs` input channels x all taps.
// Blocks that share a weight slice are placed on one XCD (blockIdx % 8) so the slice leaves HBM / Infinity Cache once.
#include <type_traits>
#include "vt_common.h"
#include "vt_uconv.h"
#include "vt_prof.h"

// weight fragments: A/B of the nt cache policy (-DVLATOUCH_UCONV_NT; each fragment is read by the m-tiles of one launch only)
#ifdef VLATOUCH_UCONV_NT
#define VT_UCONV_WLD(ptr) __builtin_nontemporal_load(ptr)
#else
#define VT_UCONV_WLD(ptr) (*(ptr))
#endif

namespace {

constexpr int WD = 8;   // weight k-steps in flight per wave (8 x 2 KiB)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4_t ldv(const float* p) { return *reinterpret_cast<const float4_t*>(p); }
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// sum over each aligned group of `n` lanes (n = 1, 2, 4 ... 64, wave-uniform), result in every lane: DPP inside a row of 16 (no LDS
// crossbar round trips), permlane swaps across rows
__device__ __forceinline__ float seg_sum(float v, int n) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  if (n >= 2) v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  if (n >= 4) v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  if (n >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  if (n >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
  // across rows: v_permlane16_swap pairs rows (0,1) and (2,3), v_permlane32_swap the two halves — VALU, no ds_bpermute round trip (round 5)
  if (n >= 32) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
  }
  if (n >= 64) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
  }
  return v;
}

template <int J>
__global__ __launch_bounds__(256) void uconv_kernel(const UConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  // block -> (weight slice id, m-tile): ids that differ only in the m-tile share blockIdx % 8
  const int rest = blockIdx.x >> 3;
  const int mt = rest % p.mtiles;
  int wid = (rest / p.mtiles) * 8 + (blockIdx.x & 7);
  if (wid >= p.nw) return;
  const int slice = wid % p.S; wid /= p.S;
  const int ntall = p.ntiles * (1 + p.has_res);
  const int nta = wid % ntall; wid /= ntall;
  const int par = wid % p.npar;
  const int net = wid / p.npar;
  const bool is_res = nta >= p.ntiles;                // the 1x1 residual convolution: its own n-tiles over the same resolved input
  const int nt = is_res ? nta - p.ntiles : nta;

  long long* tb = (p.tbuf && blockIdx.x < 2048 && tid == 0) ? p.tbuf + (long)blockIdx.x * 8 : nullptr;
  if (tb) tb[0] = wall_clock64();
  const int cs = p.cs, kcs = cs >> 5;
  const int ntt = is_res ? 1 : p.ntaps;
  const int nsteps = kcs * ntt;
  const short8_t* wbase = reinterpret_cast<const short8_t*>(is_res ? p.Wr + (long)net * p.wr_gs : p.Wp + (long)net * p.w_gs + (long)par * p.w_ps) +
                          ((long)((nt * 4 + wave) * p.nc32 + slice * kcs) * ntt) * 128 + lane;
  short8_t wh[WD], wl[WD];
#pragma unroll
  for (int d = 0; d < WD; ++d) {
    const short8_t* q = wbase + (long)min(d, nsteps - 1) * 128;
    wh[d] = VT_UCONV_WLD(q); wl[d] = VT_UCONV_WLD(q + 64);
  }

  // ------------------------------------------------------------------ prologue: resolve the input slice
  const int c_abs = slice * cs;
  const USrc& S = c_abs >= p.c_split ? p.src[1] : p.src[0];    // by reference: the 200-byte descriptor copied into SGPRs cost 62 spills to VGPR lanes; the fields it needs are scalar loads from the kernarg segment
  const int c0 = c_abs >= p.c_split ? c_abs - p.c_split : c_abs;
  const int Tin = p.Tin, c4n = cs >> 2;
  const int c4sh = __builtin_ctz(c4n);                                    // cs = 32 << i
  // row -> (sample, tick) for ANY Tin <= 64 (48-tick chunks: Tin = 48 / 24 / 12): r / Tin as a multiply-shift, exact for r < 128 rows per block
  // (magic = ceil(2^16 / Tin): the error term r * (magic * Tin - 2^16) / (Tin * 2^16) < 128 / 2^16 < 1 / Tin)
  const int tmagic = (65536 + Tin - 1) / Tin;
  const int rows_in = p.nsamp * Tin;
  const int total4 = rows_in * c4n;
  const int b0 = mt * p.nsamp;
  float4* stage = reinterpret_cast<float4*>(smem);
  float4* rstage = reinterpret_cast<float4*>(smem + p.lds_rstage);
  float2* stats = reinterpret_cast<float2*>(smem + p.lds_stats);
  char* hiP = smem + p.lds_hi;
  char* loP = smem + p.lds_lo;
  const int pitch = p.pitch;
  const float* sp = S.p + (long)net * S.gs;
  const float* sbias = S.bias ? S.bias + (long)net * S.vec_gs : nullptr;
  const float* rp = S.res_mode ? S.res + (long)net * S.res_gs : nullptr;
  const float* rbias = (S.res_mode == 2 && S.res_bias) ? S.res_bias + (long)net * S.vec_gs : nullptr;

  // GroupNorm gains / offsets and the FiLM rows of this slice: loaded into registers BEFORE the gather (their latency hides behind it),
  // parked in LDS after it.  Layout (float4 units): gamma[c4n] | beta[c4n] | per sample: scale[c4n] | bias[c4n].
  const int npar4 = S.cpg > 0 ? (2 + (S.film_s ? 2 * p.nsamp : 0)) * c4n : 0;
  float4_t pv[3];
  {
    const float* gam = S.cpg > 0 ? S.gamma + (long)net * S.vec_gs + c0 : nullptr;
    const float* bet = S.cpg > 0 ? S.beta + (long)net * S.vec_gs + c0 : nullptr;
    const float* fs = S.film_s ? S.film_s + (long)net * S.film_s_gs + S.film_off + c0 : nullptr;
    const float* fcb = S.film_s ? S.film_c + (long)net * S.film_c_gs + S.film_off + c0 : nullptr;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = tid + 256 * k;
      pv[k] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (i < npar4) {
        const int row = i >> c4sh, c4 = i & (c4n - 1);
        if (row == 0) pv[k] = ldv(gam + c4 * 4);
        else if (row == 1) pv[k] = ldv(bet + c4 * 4);
        else {
          const int samp = (row - 2) >> 1, which = (row - 2) & 1;
          const int b = min(b0 + samp, p.B - 1);
          pv[k] = ldv(fs + which * S.film_C + c4 * 4) + ldv(fcb + (long)b * S.film_ld + which * S.film_C + c4 * 4);
        }
      }
    }
  }
  // identity residual (a plain tensor): at most 4 elements per thread, loaded now, parked in LDS after the gather
  float4_t rv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    rv[k] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const int e = tid + 256 * k;
    if (S.res_mode == 1 && e < total4) {
      const int r = e >> c4sh, c4 = e & (c4n - 1);
      const int samp = (r * tmagic) >> 16, t = r - samp * Tin;
      if (b0 + samp < p.B) rv[k] = ldv(rp + ((long)(b0 + samp) * Tin + t) * S.res_ld + c0 + c4 * 4);
    }
  }
  // Every value is a sum over the producer's slabs: a latency chain unless the loads of a thread are issued together.  16 loads per
  // round: 4 elements x 4 slabs when a thread owns several elements, 1 element x 16 slabs otherwise; addresses clamped and results
  // zeroed by select so that no load sits behind a branch.  Slabs are added in index order after the bias.
  auto gather = [&](const float* gp, long gld, int ns, long gslab, const float* gbias, float4_t* dst) {
    const float4_t z4 = {0.f, 0.f, 0.f, 0.f};
    // element e of the slice -> offset of its 4 channels in the source (0 when the row lies beyond the batch), channel of the bias
    auto locate = [&](int e, bool& ok, int& c) -> long {
      const int r = e >> c4sh, c4 = e & (c4n - 1);
      const int samp = (r * tmagic) >> 16, t = r - samp * Tin;
      ok = e < total4 && b0 + samp < p.B;
      c = ok ? c0 + c4 * 4 : c0;
      return ok ? ((long)(b0 + samp) * Tin + t) * gld + c : 0L;
    };
    if (total4 > 512) {
      for (int e0 = tid; e0 < total4; e0 += 1024) {
        bool k0, k1, k2, k3; int c_0, c_1, c_2, c_3;
        const long o0 = locate(e0, k0, c_0), o1 = locate(e0 + 256, k1, c_1), o2 = locate(e0 + 512, k2, c_2), o3 = locate(e0 + 768, k3, c_3);
        float4_t v0 = gbias ? ldv(gbias + c_0) : z4, v1 = gbias ? ldv(gbias + c_1) : z4, v2 = gbias ? ldv(gbias + c_2) : z4, v3 = gbias ? ldv(gbias + c_3) : z4;
        for (int s0 = 0; s0 < ns; s0 += 4) {
          const long q0 = (long)min(s0, ns - 1) * gslab, q1 = (long)min(s0 + 1, ns - 1) * gslab, q2 = (long)min(s0 + 2, ns - 1) * gslab, q3 = (long)min(s0 + 3, ns - 1) * gslab;
          const float4_t a0 = ldv(gp + o0 + q0), b0_ = ldv(gp + o1 + q0), c0_ = ldv(gp + o2 + q0), d0 = ldv(gp + o3 + q0);
          const float4_t a1 = ldv(gp + o0 + q1), b1_ = ldv(gp + o1 + q1), c1_ = ldv(gp + o2 + q1), d1 = ldv(gp + o3 + q1);
          const float4_t a2 = ldv(gp + o0 + q2), b2_ = ldv(gp + o1 + q2), c2_ = ldv(gp + o2 + q2), d2 = ldv(gp + o3 + q2);
          const float4_t a3 = ldv(gp + o0 + q3), b3_ = ldv(gp + o1 + q3), c3_ = ldv(gp + o2 + q3), d3 = ldv(gp + o3 + q3);
          v0 += a0; v1 += b0_; v2 += c0_; v3 += d0;
          if (s0 + 1 < ns) { v0 += a1; v1 += b1_; v2 += c1_; v3 += d1; }
          if (s0 + 2 < ns) { v0 += a2; v1 += b2_; v2 += c2_; v3 += d2; }
          if (s0 + 3 < ns) { v0 += a3; v1 += b3_; v2 += c3_; v3 += d3; }
        }
        if (e0 < total4) dst[e0] = k0 ? v0 : z4;
        if (e0 + 256 < total4) dst[e0 + 256] = k1 ? v1 : z4;
        if (e0 + 512 < total4) dst[e0 + 512] = k2 ? v2 : z4;
        if (e0 + 768 < total4) dst[e0 + 768] = k3 ? v3 : z4;
      }
    } else {
      for (int e = tid; e < total4; e += 256) {
        bool ok; int c;
        const long o = locate(e, ok, c);
        float4_t v = gbias ? ldv(gbias + c) : z4;
        for (int s0 = 0; s0 < ns; s0 += 8) {
          const float4_t t0 = ldv(gp + o + (long)min(s0, ns - 1) * gslab), t1 = ldv(gp + o + (long)min(s0 + 1, ns - 1) * gslab);
          const float4_t t2 = ldv(gp + o + (long)min(s0 + 2, ns - 1) * gslab), t3 = ldv(gp + o + (long)min(s0 + 3, ns - 1) * gslab);
          const float4_t t4 = ldv(gp + o + (long)min(s0 + 4, ns - 1) * gslab), t5 = ldv(gp + o + (long)min(s0 + 5, ns - 1) * gslab);
          const float4_t t6 = ldv(gp + o + (long)min(s0 + 6, ns - 1) * gslab), t7 = ldv(gp + o + (long)min(s0 + 7, ns - 1) * gslab);
          v += t0;
          if (s0 + 1 < ns) v += t1;
          if (s0 + 2 < ns) v += t2;
          if (s0 + 3 < ns) v += t3;
          if (s0 + 4 < ns) v += t4;
          if (s0 + 5 < ns) v += t5;
          if (s0 + 6 < ns) v += t6;
          if (s0 + 7 < ns) v += t7;
        }
        dst[e] = ok ? v : z4;
      }
    }
  };
  if (S.nslabs == 0 && ((S.ld & 3) || S.cvalid < S.C)) {        // the sampler state itself: [B][T][10] rows, channels >= cvalid are zero
    for (int e = tid; e < total4; e += 256) {
      const int r = e >> c4sh, c4 = e & (c4n - 1);
      const int samp = (r * tmagic) >> 16, t = r - samp * Tin;
      const int b = b0 + samp, c = c0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < p.B) {
        const float* q = sp + ((long)b * Tin + t) * S.ld + c;
        v.x = c + 0 < S.cvalid ? q[0] : 0.f; v.y = c + 1 < S.cvalid ? q[1] : 0.f;
        v.z = c + 2 < S.cvalid ? q[2] : 0.f; v.w = c + 3 < S.cvalid ? q[3] : 0.f;
      }
      stage[e] = v;
    }
  } else {
    gather(sp, S.ld, S.nslabs ? S.nslabs : 1, S.slab, S.nslabs ? sbias : nullptr, reinterpret_cast<float4_t*>(stage));
  }
  if (S.res_mode == 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (tid + 256 * k < total4) reinterpret_cast<float4_t*>(rstage)[tid + 256 * k] = rv[k];
  } else if (S.res_mode == 2) gather(rp, S.res_ld, S.res_nslabs, S.res_slab, rbias, reinterpret_cast<float4_t*>(rstage));
  if (tb) tb[1] = wall_clock64();      // gathers done (their loads waited for)
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (tid + 256 * k < npar4) reinterpret_cast<float4_t*>(smem + p.lds_par)[tid + 256 * k] = pv[k];
  for (int i = tid + 768; i < npar4; i += 256) {          // (not reached with the tile shapes the driver picks: <= 768 parameter float4s)
    const int row = i >> c4sh, c4 = i & (c4n - 1);
    float4_t v;
    if (row == 0) v = ldv(S.gamma + (long)net * S.vec_gs + c0 + c4 * 4);
    else if (row == 1) v = ldv(S.beta + (long)net * S.vec_gs + c0 + c4 * 4);
    else {
      const int samp = (row - 2) >> 1, which = (row - 2) & 1;
      const int b = min(b0 + samp, p.B - 1);
      v = ldv(S.film_s + (long)net * S.film_s_gs + S.film_off + c0 + which * S.film_C + c4 * 4) +
          ldv(S.film_c + (long)net * S.film_c_gs + S.film_off + c0 + (long)b * S.film_ld + which * S.film_C + c4 * 4);
    }
    reinterpret_cast<float4_t*>(smem + p.lds_par)[i] = v;
  }
  // zero halo rows (two above, two below each sample) of both operand planes
  {
    const int p16 = pitch >> 4;
    const int nz = p.nsamp * 4 * p16;
    for (int e = tid; e < nz; e += 256) {
      const int rr = e / p16, ch = e - rr * p16;
      const int samp = rr >> 2, h = rr & 3;
      const int row = samp * (Tin + 4) + (h < 2 ? h : Tin + h);
      *reinterpret_cast<uint4*>(hiP + row * pitch + ch * 16) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(loP + row * pitch + ch * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  if (tb) tb[2] = wall_clock64();
  if (S.cpg > 0) {   // GroupNorm statistics: unit = (sample, group); mean, then centred sum of squares, the unit's values held in registers
    // cpg, cs (hence groups per slice), the threads per unit are powers of two: shifts, no integer division in this phase
    const int cpg4 = S.cpg >> 2, sh = __builtin_ctz(cpg4), gsh = __builtin_ctz(cs) - __builtin_ctz(S.cpg), gps = 1 << gsh;
    const int units = p.nsamp * gps;
    int tpu = 64, tsh6 = 6;
    while (tpu > 1 && tpu * units > 256) { tpu >>= 1; --tsh6; }
    const int upp = 256 >> tsh6;                     // units per pass
    const int n4 = Tin * cpg4;                       // float4s per unit
    const float inv_n = 1.0f / (float)(n4 * 4);
    const int li = tid & (tpu - 1);
    const float4_t* st4 = reinterpret_cast<const float4_t*>(stage);
    for (int u0 = 0; u0 < units; u0 += upp) {
      const int u = u0 + (tid >> tsh6);
      const bool live = u < units;
      const int samp = live ? u >> gsh : 0, gg = live ? u & (gps - 1) : 0;
      const float4_t* base = st4 + (samp * Tin) * c4n + gg * cpg4;
      float4_t xv[8];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = li + k * tpu;
        xv[k] = (live && i < n4) ? base[(i >> sh) * c4n + (i & (cpg4 - 1))] : (float4_t){0.f, 0.f, 0.f, 0.f};
        s += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
      }
      for (int i = li + 8 * tpu; live && i < n4; i += tpu) { const float4_t x = base[(i >> sh) * c4n + (i & (cpg4 - 1))]; s += (x[0] + x[1]) + (x[2] + x[3]); }
      s = seg_sum(s, tpu);
      const float mean = s * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (live && li + k * tpu < n4) {
          const float4_t d = xv[k] - mean;
          q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
      }
      for (int i = li + 8 * tpu; live && i < n4; i += tpu) {
        const float4_t d = base[(i >> sh) * c4n + (i & (cpg4 - 1))] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
      q = seg_sum(q, tpu);
      if (live && li == 0) stats[u] = make_float2(mean, rsqrtf(q * inv_n + p.eps));
    }
    __syncthreads();
  }
  if (tb) tb[3] = wall_clock64();
  {
    float* mat = (S.mat && nta == 0 && par == 0) ? S.mat + (long)net * S.mat_gs : nullptr;
    const int gps = S.cpg > 0 ? cs / S.cpg : 1, cpgsh = S.cpg > 0 ? __builtin_ctz(S.cpg) : 0;
    const float4_t* par4 = reinterpret_cast<const float4_t*>(smem + p.lds_par);    // gamma | beta | per sample: FiLM scale | FiLM bias
    for (int e = tid; e < total4; e += 256) {
      const int r = e >> c4sh, c4 = e & (c4n - 1);
      const int samp = (r * tmagic) >> 16, t = r - samp * Tin;
      const int b = b0 + samp;
      const int c = c0 + c4 * 4;
      float4_t y = reinterpret_cast<const float4_t*>(stage)[e];
      if (b < p.B) {
        if (S.cpg > 0) {
          const float2 st = stats[samp * gps + ((c4 * 4) >> cpgsh)];
          y = (y - st.x) * st.y * par4[c4] + par4[c4n + c4];
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = act_apply(y[i], VT_ACT_MISH);
          if (S.film_s) y = par4[(2 + 2 * samp) * c4n + c4] * y + par4[(3 + 2 * samp) * c4n + c4];
        }
        if (S.res_mode) y += reinterpret_cast<const float4_t*>(rstage)[e];
        if (mat) *reinterpret_cast<float4_t*>(mat + ((long)b * Tin + t) * S.mat_ld + c) = y;
      }
      const uint32_t h0 = pk_bf16(y[0], y[1]), h1 = pk_bf16(y[2], y[3]);
      const uint32_t l0 = pk_bf16(y[0] - __uint_as_float(h0 << 16), y[1] - __uint_as_float(h0 & 0xffff0000u));
      const uint32_t l1 = pk_bf16(y[2] - __uint_as_float(h1 << 16), y[3] - __uint_as_float(h1 & 0xffff0000u));
      const int off = (samp * (Tin + 4) + 2 + t) * pitch + c4 * 8;
      *reinterpret_cast<uint2*>(hiP + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(loP + off) = make_uint2(l0, l1);
    }
  }
  __syncthreads();

  if (tb) tb[4] = wall_clock64();
  // ------------------------------------------------------------------ k-loop (no barrier: LDS is read-only from here on)
  int abase[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int m = j * 16 + l15;
    int samp = m / p.Tq, t = m - samp * p.Tq;
    if (samp >= p.nsamp) { samp = 0; t = 0; }        // padding rows of a block whose 16 J rows are not whole samples (Tq = 12: 2 x 12 of 32): computed, never stored
    abase[j] = (samp * (Tin + 4) + 2 + t * p.stride) * pitch + g * 16;
  }
  float4_t acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
  const int* offs = p.off + par * 6;
  const int zoff = is_res ? 5 : 0;                    // slot 5 of the offset table = 0: the 1x1 convolution reads the row itself
  // one k-step: 32 input channels of one tap.  (kc, tap) advance tap-first, the order of the packed weight stream.
  int kc = 0, tap = 0;
  auto step = [&](const short8_t& whs, const short8_t& wls) {
    const int aoff = offs[zoff + tap] * pitch + kc * 64;
    const bf16x8_t whv = __builtin_bit_cast(bf16x8_t, whs), wlv = __builtin_bit_cast(bf16x8_t, wls);
    bf16x8_t ah[J], al[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      ah[j] = *reinterpret_cast<const bf16x8_t*>(hiP + abase[j] + aoff);
      al[j] = *reinterpret_cast<const bf16x8_t*>(loP + abase[j] + aoff);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {                     // small terms first, as vt_gemm.hip's split mode orders them
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlv, ah[j], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whv, al[j], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whv, ah[j], acc[j], 0, 0, 0);
    }
    ++tap;
    if (tap == ntt) { tap = 0; ++kc; }
  };
  // full groups of WD steps: branch-free, every slot reloaded (index clamped past the end) so the number of loads in flight is constant
  // and hipcc's counted vmcnt waits stay exact (a guarded reload made it drain the queue at every step); then the tail without loads
  int s0 = 0;
  for (; s0 + WD <= nsteps; s0 += WD) {
#pragma unroll
    for (int d = 0; d < WD; ++d) {
      step(wh[d], wl[d]);
      const short8_t* q = wbase + (long)min(s0 + d + WD, nsteps - 1) * 128;
      wh[d] = VT_UCONV_WLD(q); wl[d] = VT_UCONV_WLD(q + 64);
    }
  }
#pragma unroll
  for (int d = 0; d < WD; ++d)
    if (s0 + d < nsteps) step(wh[d], wl[d]);

  if (tb) { asm volatile("" : "+v"(acc[0])); tb[5] = wall_clock64(); }
  // ------------------------------------------------------------------ epilogue: lane holds rows m = j*16 + l15, channels n0 + g*4 .. +3
  const int n = nt * 64 + wave * 16 + g * 4;
  float* ob = (is_res ? p.rout + (long)net * p.rout_gs + (long)slice * p.rout_slab : p.out + (long)net * p.out_gs + (long)slice * p.out_slab) + n;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int m = j * 16 + l15;
    const int samp = m / p.Tq, t = m - samp * p.Tq;
    const int b = b0 + samp;
    if (b >= p.B || samp >= p.nsamp) continue;
    const long orow = ((long)b * p.Tq + t) * p.omul + par;
    *reinterpret_cast<float4*>(ob + orow * p.ldc) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
  }
  if (tb) { tb[6] = wall_clock64(); tb[7] = (long long)((nta << 20) | (slice << 8) | J); }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// final_conv.0's GroupNorm + Mish, final_conv.1 (Conv1d(C, dim, 1)) of both nets and the Euler-Maruyama update (bridge_model.py:363-385):
// one block of 16 waves per sample, 8 waves per net.  The 1x1 convolution is exact fp32 on the vector ALU (C*dim*T FMAs per net) with its
// weights staged in LDS; every global load of a phase is issued before the first use.
__global__ __launch_bounds__(1024) void ufinal_kernel(const UFinalParams p) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];      // [2 nets][T*C activations | dim*C weights]
  __shared__ float2 stats[2][64];
  __shared__ float outv[2][64 * 16];
  __shared__ float outb[2][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = wv >> 3, w8 = wv & 7, t512 = tid & 511;
  const int b = blockIdx.x;
  const int T = p.T, C = p.C, c4n = C >> 2, total4 = T * c4n;
  const int ngr = C / p.cpg;
  const bool live = net < p.nets;
  float* act = fsm + (size_t)net * (T * C + p.dim * C);
  float* wl = act + T * C;
  if (live) {
    const float* sp = p.slabs + (long)net * p.gs + (long)b * T * C;
    const float* bias = p.bias + (long)net * p.vec_gs;
    const float* W = p.out_w + (long)net * p.ow_gs;
    for (int e = t512; e < p.dim * c4n; e += 512) reinterpret_cast<float4*>(wl)[e] = ld4(W + e * 4);
    if (t512 < p.dim) outb[net][t512] = p.out_b[(long)net * p.ob_gs + t512];
    for (int e0 = t512; e0 < total4; e0 += 2048) {           // 4 elements x 4 slabs in flight
      float4 v0, v1, v2, v3;
      const int ea = min(e0, total4 - 1), eb = min(e0 + 512, total4 - 1), ec = min(e0 + 1024, total4 - 1), ed = min(e0 + 1536, total4 - 1);
      const int oa = (ea / c4n) * C + (ea % c4n) * 4, ob = (eb / c4n) * C + (eb % c4n) * 4;
      const int oc = (ec / c4n) * C + (ec % c4n) * 4, od = (ed / c4n) * C + (ed % c4n) * 4;
      v0 = ld4(bias + (ea % c4n) * 4); v1 = ld4(bias + (eb % c4n) * 4); v2 = ld4(bias + (ec % c4n) * 4); v3 = ld4(bias + (ed % c4n) * 4);
      for (int s0 = 0; s0 < p.nslabs; s0 += 4) {
        float4 ta[4], tb[4], tc[4], td[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long so = (long)min(s0 + u, p.nslabs - 1) * p.slab;
          ta[u] = ld4(sp + oa + so); tb[u] = ld4(sp + ob + so); tc[u] = ld4(sp + oc + so); td[u] = ld4(sp + od + so);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s0 + u < p.nslabs) { add4(v0, ta[u]); add4(v1, tb[u]); add4(v2, tc[u]); add4(v3, td[u]); }
      }
      if (e0 < total4) *reinterpret_cast<float4*>(act + oa) = v0;
      if (e0 + 512 < total4) *reinterpret_cast<float4*>(act + ob) = v1;
      if (e0 + 1024 < total4) *reinterpret_cast<float4*>(act + oc) = v2;
      if (e0 + 1536 < total4) *reinterpret_cast<float4*>(act + od) = v3;
    }
  }
  __syncthreads();
  if (live) {   // group statistics: a wave per group (mean, then centred sum of squares from the same registers)
    const int cpg4 = p.cpg >> 2, sh = __builtin_ctz(cpg4);
    const int n4 = T * cpg4;
    const float inv_n = 1.0f / (float)(n4 * 4);
    const float4_t* a4 = reinterpret_cast<const float4_t*>(act);
    for (int u = w8; u < ngr; u += 8) {
      const float4_t* base = a4 + u * cpg4;
      float4_t xv[8];
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = lane + 64 * k;
        xv[k] = i < n4 ? base[(i >> sh) * c4n + (i & (cpg4 - 1))] : (float4_t){0.f, 0.f, 0.f, 0.f};
        s += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
      }
      for (int i = lane + 512; i < n4; i += 64) { const float4_t x = base[(i >> sh) * c4n + (i & (cpg4 - 1))]; s += (x[0] + x[1]) + (x[2] + x[3]); }
      const float mean = wave_sum(s) * inv_n;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (lane + 64 * k < n4) { const float4_t d = xv[k] - mean; q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]); }
      for (int i = lane + 512; i < n4; i += 64) { const float4_t d = base[(i >> sh) * c4n + (i & (cpg4 - 1))] - mean; q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]); }
      q = wave_sum(q);
      if (lane == 0) stats[net][u] = make_float2(mean, rsqrtf(q * inv_n + p.gn_eps));
    }
  }
  __syncthreads();
  if (live) {
    const float* gam = p.gamma + (long)net * p.vec_gs;
    const float* bet = p.beta + (long)net * p.vec_gs;
    for (int e = t512; e < total4; e += 512) {
      const int t = e / c4n, c = (e - t * c4n) * 4;
      const float2 st = stats[net][c / p.cpg];
      const float4 g4 = ld4(gam + c), b4 = ld4(bet + c);
      float4 v = *reinterpret_cast<float4*>(act + t * C + c);
      v.x = act_apply((v.x - st.x) * st.y * g4.x + b4.x, VT_ACT_MISH); v.y = act_apply((v.y - st.x) * st.y * g4.y + b4.y, VT_ACT_MISH);
      v.z = act_apply((v.z - st.x) * st.y * g4.z + b4.z, VT_ACT_MISH); v.w = act_apply((v.w - st.x) * st.y * g4.w + b4.w, VT_ACT_MISH);
      *reinterpret_cast<float4*>(act + t * C + c) = v;
    }
  }
  __syncthreads();
  if (live) {   // out[t][d] = sum_c act[t][c] W[d][c] + b[d]: a wave per row, lanes over channels; the dim partial sums reduce together
    for (int t = w8; t < T; t += 8) {              // (one butterfly of 6 steps over 16 independent values, not 16 dependent butterflies)
      float s[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) s[d] = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float a = act[t * C + c];
#pragma unroll
        for (int d = 0; d < 16; ++d) if (d < p.dim) s[d] += a * wl[d * C + c];
      }
#pragma unroll
      for (int d = 0; d < 16; ++d) s[d] = wave_sum(s[d]);            // DPP + permlane swaps (vt_common.h)
#pragma unroll
      for (int d = 0; d < 16; ++d) if (lane == d && d < p.dim) outv[net][t * 16 + d] = s[d] + outb[net][d];
    }
  }
  __syncthreads();
  const int n = T * p.dim;
  for (int e = tid; e < n; e += 1024) {
    const int t = e / p.dim, d = e - t * p.dim;
    const long i = (long)b * n + e;
    if (p.vs)
      for (int k = 0; k < p.nets; ++k) p.vs[(long)k * p.B * n + i] = outv[k][t * 16 + d];
    if (p.do_sde) {
      const float sv = outv[1][t * 16 + d] * p.gi;
      const float bb = outv[0][t * 16 + d] - p.gdg * sv * p.eps_t;
      float xn = p.backward ? p.x[i] - (bb - p.score_eps * sv) * p.dt : p.x[i] + (bb + p.score_eps * sv) * p.dt;
      if (p.z) xn += p.noise_scale * (p.d * p.z[i]);
      p.x[i] = xn;
      if (p.traj) p.traj[i] = xn;
    }
  }
}

__global__ void uconv_pack_kernel(const float* __restrict__ Wm, uint16_t* __restrict__ out, int nets, int N, int ntaps, int cinp, int nc32, long out_gs) {
  const int ntt = ntaps;
  const long per_net = (long)(N / 16) * nc32 * ntt * 64;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_net * nets) return;
  const int net = (int)(i / per_net);
  long r = i - (long)net * per_net;
  const int lane = (int)(r & 63); r >>= 6;
  const int tap = (int)(r % ntt); r /= ntt;
  const int c32 = (int)(r % nc32); r /= nc32;
  const int n16 = (int)r;                             // = nt*4 + wave
  const int nrow = n16 * 16 + (lane & 15);
  const int c = c32 * 32 + (lane >> 4) * 8;
  float w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int cc = c + k;
    w[k] = cc < cinp ? Wm[((long)net * N + nrow) * ((long)ntaps * cinp) + (long)tap * cinp + cc] : 0.f;
  }
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pk_bf16(w[2 * j], w[2 * j + 1]);
    l[j] = pk_bf16(w[2 * j] - __uint_as_float(h[j] << 16), w[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
  }
  uint16_t* o = out + (long)net * out_gs + (((long)n16 * nc32 + c32) * ntt + tap) * 1024 + lane * 8;
  *reinterpret_cast<uint4*>(o) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(o + 512) = make_uint4(l[0], l[1], l[2], l[3]);
}

struct USinArgs { float t[64]; };
__global__ void usin_kernel(const USinArgs a, int n, float* __restrict__ out, int dsed) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * dsed) return;
  const int k = i / dsed, c = i - k * dsed;
  const int half = dsed >> 1;
  const int j = c < half ? c : c - half;
  const float f = expf((float)j * -(9.210340371976184f / ((float)half - 1.0f)));     // conditional_unet_1D.py:15-16
  const float ang = a.t[k] * f;
  out[i] = c < half ? sinf(ang) : cosf(ang);
}

}  // namespace

static long long* g_tbuf = nullptr;
static int g_tmax = 0, g_tidx = 0;
extern "C" int vt_uconv_set_timing(long long* buf, int max_launches) { g_tbuf = buf; g_tmax = max_launches; g_tidx = 0; return VT_OK; }

int vt_uconv_launch(const UConvParams& p_in, int J, size_t lds_bytes, hipStream_t s) {
  UConvParams p = p_in;
  p.tbuf = (g_tbuf && g_tidx < g_tmax) ? g_tbuf + (long)(g_tidx++) * 2048 * 8 : nullptr;
  static const bool attr_set = [] {      // tiles may use more than the default 64 KiB of dynamic LDS
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&uconv_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
    ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&uconv_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&uconv_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&uconv_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess && ok;
    return ok;
  }();
  if (!attr_set || lds_bytes > 160 * 1024) return VT_ERR_LAUNCH;
  const int nw8 = (p.nw + 7) / 8 * 8;
  dim3 grid(nw8 * p.mtiles);
  // vt_prof class 6: MFMA flops actually issued (3 bf16 products per fp32 product) and the bytes a launch must move at least once
  // (its weights + the slabs it resolves + the slabs it writes)
  const double rows = (double)p.B * p.Tq * p.npar, K = (double)p.nc32 * 32 * p.ntaps, Cin = (double)p.nc32 * 32;
  const double flops = 3.0 * 2.0 * rows * p.N * (K + (p.has_res ? Cin : 0.0)) * p.nets;
  const double bytes = ((double)p.N * (K + (p.has_res ? Cin : 0.0)) * 4.0 + (double)p.B * p.Tin * Cin * 4.0 * (p_in.src[0].nslabs > 0 ? p_in.src[0].nslabs : 1) +
                        rows * p.N * 4.0 * p.S * (1 + p.has_res)) * p.nets;
  VtProfScope prof(6, flops, bytes, s);
  if (J == 4) hipLaunchKernelGGL((uconv_kernel<4>), grid, dim3(256), lds_bytes, s, p);
  else if (J == 3) hipLaunchKernelGGL((uconv_kernel<3>), grid, dim3(256), lds_bytes, s, p);     // 48 rows: chunks whose levels are multiples of 3 ticks (T = 48 / 24 / 12)
  else if (J == 2) hipLaunchKernelGGL((uconv_kernel<2>), grid, dim3(256), lds_bytes, s, p);
  else if (J == 1) hipLaunchKernelGGL((uconv_kernel<1>), grid, dim3(256), lds_bytes, s, p);
  else return VT_ERR_ARG;
  return vt_check_launch();
}

// dynamic + static LDS of the final kernel (activations and the 1x1 weights of both nets + its small static arrays): must fit the CU's 160 KiB
size_t vt_ufinal_lds_bytes(int T, int dim, int C) { return (size_t)2 * (T + dim) * C * 4 + 2 * 64 * 8 + 2 * 64 * 16 * 4 + 2 * 16 * 4; }

int vt_ufinal_launch(const UFinalParams& p, hipStream_t s) {
  if (p.T > 64 || p.dim > 16 || p.nets > 2 || (p.C & 3) || p.C / p.cpg > 64) return VT_ERR_UNSUPPORTED;
  if (vt_ufinal_lds_bytes(p.T, p.dim, p.C) > 160 * 1024) return VT_ERR_UNSUPPORTED;
  static const bool attr_set = hipFuncSetAttribute(reinterpret_cast<const void*>(&ufinal_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
  if (!attr_set) return VT_ERR_LAUNCH;
  hipLaunchKernelGGL(ufinal_kernel, dim3(p.B), dim3(1024), (size_t)2 * (p.T + p.dim) * p.C * 4, s, p);
  return vt_check_launch();
}

int vt_uconv_pack(const float* Wm, uint16_t* out, int nets, int N, int ntaps, int cinp, int nc32, long out_gs, hipStream_t s) {
  const long total = (long)nets * (N / 16) * nc32 * ntaps * 64;
  hipLaunchKernelGGL(uconv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Wm, out, nets, N, ntaps, cinp, nc32, out_gs);
  return vt_check_launch();
}

int vt_usin_launch(const float* ts_host, int n, float* out, int dsed, hipStream_t s) {
  if (n > 64) return VT_ERR_UNSUPPORTED;
  USinArgs a;
  for (int i = 0; i < 64; ++i) a.t[i] = i < n ? ts_host[i] : 0.f;
  hipLaunchKernelGGL(usin_kernel, dim3((n * dsed + 255) / 256), dim3(256), 0, s, a, n, out, dsed);
  return vt_check_launch();
}
