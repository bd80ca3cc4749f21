// vt_attn_kvt.hip — cross-attention against a CACHED condition (RDT; 16-bit: bf16 or IEEE fp16, template parameter T).  The cache is a per-head TILE STREAM over the rows
// m = b*L + l of the whole batch (samples back to back):
//   tile(h, t = m/64) at KV + (h*T + t) * 8192 elements = [K: 64 rows x 64 d (after k_norm)][Vt: 64 d x 64 rows]
//   (Vt keys in MFMA k order: row kk of the tile sits at position vt_kpos(kk), see vt_kernels.h)
// i.e. 16 KiB contiguous per 64 keys, written once per chunk straight from the K / V projection GEMM epilogues (whose 32-row
// patches are exactly half tiles: aligned 8-byte stores, no per-sample bookkeeping in the GEMM) or by the retile kernels below
// for small shapes, and re-read by every denoise step.  Sample b attends to rows [b*L, (b+1)*L): its first and last tile may
// also hold a neighbour's keys (or never-written rows past the end), which are masked in the softmax and zeroed in the Vt
// fragment (0 * garbage could be NaN).  Both halves of a tile are plain 128-byte rows, so a tile goes HBM -> LDS by DMA
// (global_load_lds_dwordx4, source-side XOR swizzle) into a 2-stage ring while the previous tile's MFMAs run: the kernel streams
// the 1.15 GB of RDT-1B image K/V per call in whole 16-KiB bursts (vt_attn.hip, register-staged, still serves self-attention).
// Block = NW waves = 16*NW query rows of one (batch, head); fragment conventions as vt_attn.hip.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_prof.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int KT = 64;                 // keys per tile
typedef __attribute__((ext_vector_type(2))) _Float16 kvt_half2_t;
template <typename T> __device__ __forceinline__ uint32_t kvt_pk(float lo, float hi);
template <> __device__ __forceinline__ uint32_t kvt_pk<bf16_t>(float lo, float hi) { return pk_bf16(lo, hi); }
template <> __device__ __forceinline__ uint32_t kvt_pk<half_t>(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){lo, hi}, kvt_half2_t));
}
// max over the four 16-lane rows of a wave (lanes that differ in bits 4 and 5), result in every lane, on v_permlane16_swap / v_permlane32_swap
// (two VALU ops) instead of two __shfl_xor = ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ float kvt_max_rows4(float v) {
#ifdef VLATOUCH_KVT_MAX_SHFL
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
#else
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
  const unsigned x = __builtin_bit_cast(unsigned, w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
#endif
}
template <typename T> struct KvtOne;      // 1.0 in the 16-bit type (the fragment of ones that sums P on the matrix pipe)
template <> struct KvtOne<bf16_t> { static constexpr short v = 0x3f80; };
template <> struct KvtOne<half_t> { static constexpr short v = 0x3c00; };
constexpr int STAGE = 2 * KT * 128;    // K tile (64 rows x 128 B) + Vt tile (64 d-rows x 128 B) = 16 KiB

template <typename T>
__global__ __launch_bounds__(512) void attn_kvt_kernel(const VtAttnKvtParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  // parts > 1 (small batches, a single query block): blockIdx.x is the PART of the sample's key tiles this block covers; the
  // un-normalised partial (m, l, o) goes to `part_ws` and attn_combine_kernel merges the parts (flash-decoding)
  const int part = p.parts > 1 ? blockIdx.x : 0;
  const int q = (p.parts > 1 ? 0 : blockIdx.x * (nw * 16)) + wave * 16 + l15;
  const uint16_t* Q = reinterpret_cast<const uint16_t*>(p.Q) + (long)b * p.q_bs + (long)h * 64;
  const uint16_t* KV = reinterpret_cast<const uint16_t*>(p.KV) + (long)h * p.T * 8192;
  const uint8_t* km = p.kmask ? p.kmask + (long)b * p.Nk : nullptr;
  const int row0 = b * p.Nk, row1 = row0 + p.Nk;             // this sample's rows of the stream
  int t_first = row0 >> 6, t_last = (row1 - 1) >> 6;
  if (p.parts > 1) {
    const int nt = t_last - t_first + 1, per = (nt + p.parts - 1) / p.parts;
    t_first += part * per;
    t_last = min(t_last, t_first + per - 1);
  }

  Frag<T> qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    qf[ks].v = q < p.Nq ? *reinterpret_cast<const short8_t*>(Q + (long)q * p.q_rs + ks * 32 + g * 8) : (short8_t){0, 0, 0, 0, 0, 0, 0, 0};

  // DMA plan: 16 wave-instructions per tile (8 for K rows, 8 for Vt rows = 16 x 1 KiB of the contiguous tile), 4 consecutive
  // ones per wave 0..3.  lane -> (row = i*8 + lane/8, chunk position = lane%8); it fetches the chunk whose swizzled position
  // is its own.  (Streaming alone — this loop without the MFMA / softmax work — runs at ~5.6 TB/s with 2, 3, 4 or 5 stages of
  // 32 or 64 keys alike: one tile ahead already saturates what 4 blocks per CU can pull.)
  const int r_in = lane >> 3, pch = lane & 7;
  auto stage = [&](int buf, int tile) {
    char* base = smem + buf * STAGE;
    const uint16_t* src = KV + (long)tile * 8192;
    if (wave >= 4) return;                     // waves 0..3 carry 4 consecutive pieces each (measured: 20 % faster streaming
#pragma unroll                                 // than dealing the 16 pieces round-robin over all NW waves)
    for (int e = 0; e < 4; ++e) {
      const int i = wave * 4 + e;
      const int r = (i & 7) * 8 + r_in;
      const int c = pch ^ ((r >> 1) & 7);
      __builtin_amdgcn_global_load_lds((glb_void*)(src + (i * 8 + r_in) * 64 + c * 8), (lds_void*)(base + i * 1024), 16, 0, 0);
    }
  };

  float4_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float cscale = p.scale * 1.4426950408889634f;

  if (t_first <= t_last) stage(0, t_first);
  __syncthreads();
  for (int tile = t_first; tile <= t_last; ++tile) {
    const int cur = (tile - t_first) & 1;
    if (tile < t_last) stage(cur ^ 1, tile + 1);
    const char* Ks = smem + cur * STAGE;
    const char* Vs = Ks + KT * 128;
    const int key0 = tile * KT;                               // stream row of the tile's first key
    const bool partial = key0 < row0 || key0 + KT > row1;     // block-uniform

    float4_t sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag<T> kf;
        lds_frag(kf, Ks, kt * 16 + l15, ks * 4 + g);
        mma16(sacc[kt], kf, qf[ks]);
      }
    }
    // online softmax in the exp2 domain: p = exp2(s*c - m*c), c = scale*log2(e) (c > 0, so the max is taken on raw scores).
    // Masking (partial last tile, key mask) is a block-uniform slow path; full unmasked tiles pay nothing for it.
    float sv[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[kt * 4 + r] = sacc[kt][r];
    if (km || partial) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kidx = key0 + kt * 16 + g * 4 + r;
          bool ok = kidx >= row0 && kidx < row1;
          if (ok && km) ok = km[kidx - row0] != 0;
          if (!ok) sv[kt * 4 + r] = -INFINITY;
        }
    }
    float mx = sv[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new != m_run)) {          // rescale only when some row's running max moved (rare after the first tiles)
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * cscale);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      m_run = m_new;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : m_run * cscale;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -mc)); psum += sv[i]; }
    l_run += psum;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      // P fragment = this lane's own scores: k index j <-> key kb*32 + (j>>2)*16 + g*4 + (j&3)
      uint4 pw;
      pw.x = kvt_pk<T>(sv[(kb * 2) * 4 + 0], sv[(kb * 2) * 4 + 1]);
      pw.y = kvt_pk<T>(sv[(kb * 2) * 4 + 2], sv[(kb * 2) * 4 + 3]);
      pw.z = kvt_pk<T>(sv[(kb * 2 + 1) * 4 + 0], sv[(kb * 2 + 1) * 4 + 1]);
      pw.w = kvt_pk<T>(sv[(kb * 2 + 1) * 4 + 2], sv[(kb * 2 + 1) * 4 + 3]);
      Frag<T> pf;
      pf.v = __builtin_bit_cast(short8_t, pw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // the Vt tile stores its keys in the SAME k order (position g*8 + j inside each 32-key half), so the A fragment of
        // row d = dt*16 + l15 is one 16-byte chunk
        Frag<T> vf;
        lds_frag(vf, Vs, dt * 16 + l15, kb * 4 + g);
        if (partial) {          // first / last tile of the sample: rows that are not its keys must not reach the MFMA
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int kidx = key0 + kb * 32 + (j >> 2) * 16 + g * 4 + (j & 3);
            if (kidx < row0 || kidx >= row1) vf.v[j] = 0;
          }
        }
        mma16(o[dt], vf, pf);
      }
    }
    __syncthreads();      // next stage landed (the barrier drains the DMA) and this stage is free again
  }
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (p.parts > 1) {        // partial: [b][h][part][row (16*nw)][66] floats = o[64] | m | l   (m in score units, pre-scale)
    float* W = p.part_ws + ((((long)b * p.H + h) * p.parts + part) * (nw * 16) + wave * 16 + l15) * 66;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) W[dt * 16 + g * 4 + r] = o[dt][r];
    if (g == 0) { W[64] = m_run; W[65] = l; }
    return;
  }
  // a row whose probabilities sum to 0 (every key masked; or every P flushed to zero) or to inf has no softmax: zeros + the range-guard bit instead of 0 / 0 = NaN actions
  const bool l_bad = !(l > 0.f) || vt_nonfinite(l);
  const float inv = l_bad ? 0.f : 1.0f / l;
  if (l_bad && q < p.Nq) vt_range_note(p.range_flag, VT_RANGE_ATTN_EMPTY);
  if (q < p.Nq) {
    uint16_t* O = reinterpret_cast<uint16_t*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 t;
      t.x = kvt_pk<T>(o[dt][0] * inv, o[dt][1] * inv);
      t.y = kvt_pk<T>(o[dt][2] * inv, o[dt][3] * inv);
      *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = t;
    }
  }
}

// ---- ring variant: the SAME math with the tile stream staged in HALF tiles (K half, Vt half: 8 KiB each) through a ring of RING
// slots, DMA waits COUNTED (s_waitcnt vmcnt(2 * halves still wanted in flight)) and raw s_barrier, so that RING-1 half tiles stay in
// flight across the barriers instead of the single whole tile the 2-stage kernel above keeps between two draining __syncthreads():
// the kernel is bound by HBM latency x bytes in flight per CU (4 co-resident blocks; compute per tile is ~1/3 of the time a tile
// takes to arrive), and 5 x 8 KiB per block is what 4 blocks per CU can hold in 160 KiB of LDS.
// Per half h (h even: K of tile h/2, h odd: its Vt): [wait until half h landed] [barrier: everybody sees it and is done with half
// h-1] [issue half h + RING-1 into the slot of half h-1] [consume half h].
// FIXED: softmax against a FIXED maximum instead of the running one.  q and k are per-head RMS-normed (blocks.py:72-138): |q| <= 8 max|w_q|,
// |k| <= 8 max|w_k|, so |q.k| * scale <= 8 max|w_q| max|w_k| =: B, a load-time constant per layer (p.fixed_max = B).  With exp(s*scale - B)
// in (e^-2B, 1] nothing overflows and, for B <= 40, nothing underflows in fp32 or bf16: no max reduction over the 64 scores, no cross-lane
// traffic, no accumulator rescale, no data-dependent branch — 22 of the ~80 VALU instructions per 16 scores go; the row sum of P moves
// to the matrix pipe (one extra MFMA per 32 keys against a fragment of ones: it sums exactly the bf16 P that multiplies V, and arrives
// already reduced over all lanes), another 16 VALU adds.  The launcher falls back to the online form when B > 40.
template <typename T, int RING, bool FIXED, int AUX = 0>
__global__ __launch_bounds__(512) void attn_kvt_ring_kernel(const VtAttnKvtParams p) {
  constexpr int HALF = KT * 128;                     // 8 KiB
  __shared__ __attribute__((aligned(16))) char smem[RING * HALF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int part = p.parts > 1 ? blockIdx.x : 0;
  const int q = (p.parts > 1 ? 0 : blockIdx.x * (nw * 16)) + wave * 16 + l15;
  const uint16_t* Q = reinterpret_cast<const uint16_t*>(p.Q) + (long)b * p.q_bs + (long)h * 64;
  const uint16_t* KV = reinterpret_cast<const uint16_t*>(p.KV) + (long)h * p.T * 8192;
  const uint8_t* km = p.kmask ? p.kmask + (long)b * p.Nk : nullptr;
  const int row0 = b * p.Nk, row1 = row0 + p.Nk;
  int t_first = row0 >> 6, t_last = (row1 - 1) >> 6;
  if (p.parts > 1) {
    const int nt = t_last - t_first + 1, per = (nt + p.parts - 1) / p.parts;
    t_first += part * per;
    t_last = min(t_last, t_first + per - 1);
  }

  // a half tile = 8 pieces of 1 KiB (8 rows x 128 B); waves 0..3 issue 2 consecutive pieces each
  const int r_in = lane >> 3, pch = lane & 7;
  const int nh = t_first <= t_last ? 2 * (t_last - t_first + 1) : 0;
  const uint16_t* hsrc = KV + (long)t_first * 8192;    // half hh of this block's range starts at hsrc + hh * 4096
  auto stage_half = [&](int hh, int slot) {
    if (wave >= 4) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = wave * 2 + e;
      const int r = i * 8 + r_in;
      const int c = pch ^ ((r >> 1) & 7);
      __builtin_amdgcn_global_load_lds((glb_void*)(hsrc + (long)hh * 4096 + r * 64 + c * 8), (lds_void*)(smem + slot * HALF + i * 1024), 16, 0, AUX);
    }
  };
  // wait until half hh has landed (this wave's pieces), leaving the younger halves in flight; then the block-wide barrier
  auto arrive = [&](int hh) {
    const int rem = min(RING - 2, nh - 1 - hh);      // halves younger than hh that are already issued
    if (rem >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(RING >= 3 && RING <= 5, "vmcnt table above covers up to 3 younger halves");

  float4_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY;
  const float cscale = p.scale * 1.4426950408889634f;
  // FIXED: the bound is already in scaled-score units.  IEEE fp16 probabilities have 5 exponent bits: exp(s - B) in (e^-2B, 1] would sit in (and below)
  // the subnormals for every row whose largest score is well under the bound, so the fp16 form shifts the exponent up by 15 octaves (P in (2^15 e^-2B, 2^15],
  // the largest finite fp16 is 65504): the row sum l carries the same factor and O / l cancels it; the launcher admits bounds up to 10 only (P stays a normal fp16 number; 40 for bf16)
  constexpr float P_SHIFT = std::is_same<T, half_t>::value ? 15.0f : 0.0f;
  const float fixed_mc = p.fixed_max * 1.4426950408889634f - P_SHIFT;
  float4_t lacc = {0.f, 0.f, 0.f, 0.f};                           // row sums of P on the matrix pipe (every row of the tile = l)
  Frag<T> ones;
  constexpr short one16 = KvtOne<T>::v;
  ones.v = (short8_t){one16, one16, one16, one16, one16, one16, one16, one16};

#pragma unroll
  for (int hh = 0; hh < RING - 1; ++hh)
    if (hh < nh) stage_half(hh, hh);
  // the Q fragments AFTER the first halves are on their way (one memory round trip before the first MFMA instead of two: the language
  // layers' launches are one tile long), and everything retired together: an ordinary load still pending when the ring runs would make the
  // compiler drain the whole queue (vmcnt(0)) at its first use anyway
  Frag<T> qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    qf[ks].v = q < p.Nq ? *reinterpret_cast<const short8_t*>(Q + (long)q * p.q_rs + ks * 32 + g * 8) : (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(qf[ks].v));
  int slot = 0;                                       // slot of the half being consumed
  auto next_slot = [&](int s_) { return s_ + 1 == RING ? 0 : s_ + 1; };
  auto prev_slot = [&](int s_) { return s_ == 0 ? RING - 1 : s_ - 1; };

  for (int tile = t_first; tile <= t_last; ++tile) {
    const int hh = 2 * (tile - t_first);
    const int key0 = tile * KT;
    const bool partial = key0 < row0 || key0 + KT > row1;     // block-uniform
    // ---------------- K half
    arrive(hh);
    if (hh + RING - 1 < nh) stage_half(hh + RING - 1, prev_slot(slot));
    const char* Ks = smem + slot * HALF;
    float4_t sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag<T> kf;
        lds_frag(kf, Ks, kt * 16 + l15, ks * 4 + g);
        mma16(sacc[kt], kf, qf[ks]);
      }
    }
    float sv[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[kt * 4 + r] = sacc[kt][r];
    if (km || partial) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kidx = key0 + kt * 16 + g * 4 + r;
          bool ok = kidx >= row0 && kidx < row1;
          if (ok && km) ok = km[kidx - row0] != 0;
          if (!ok) sv[kt * 4 + r] = -INFINITY;
        }
    }
    if constexpr (FIXED) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -fixed_mc));      // exp2(-inf) = 0 for masked keys
    } else {
      float mx = sv[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
      mx = kvt_max_rows4(mx);
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new != m_run)) {
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * cscale);
        // the row sums live on the matrix pipe here too (round 5; below): lacc's four rows all hold l of query l15, so they take the lane's own alpha
#pragma unroll
        for (int r = 0; r < 4; ++r) lacc[r] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        m_run = m_new;
      }
      const float mc = (m_run == -INFINITY) ? 0.f : m_run * cscale;
#pragma unroll
      for (int i = 0; i < 16; ++i) sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -mc));
    }
    Frag<T> pf[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      uint4 pw;
      pw.x = kvt_pk<T>(sv[(kb * 2) * 4 + 0], sv[(kb * 2) * 4 + 1]);
      pw.y = kvt_pk<T>(sv[(kb * 2) * 4 + 2], sv[(kb * 2) * 4 + 3]);
      pw.z = kvt_pk<T>(sv[(kb * 2 + 1) * 4 + 0], sv[(kb * 2 + 1) * 4 + 1]);
      pw.w = kvt_pk<T>(sv[(kb * 2 + 1) * 4 + 2], sv[(kb * 2 + 1) * 4 + 3]);
      pf[kb].v = __builtin_bit_cast(short8_t, pw);
    }
    slot = next_slot(slot);
    // ---------------- Vt half
    arrive(hh + 1);
    if (hh + RING < nh) stage_half(hh + RING, prev_slot(slot));
    const char* Vs = smem + slot * HALF;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Frag<T> vf;
        lds_frag(vf, Vs, dt * 16 + l15, kb * 4 + g);
        if (partial) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int kidx = key0 + kb * 32 + (j >> 2) * 16 + g * 4 + (j & 3);
            if (kidx < row0 || kidx >= row1) vf.v[j] = 0;
          }
        }
        mma16(o[dt], vf, pf[kb]);
      }
    // row sums of P on the matrix pipe, in BOTH softmax forms (round 5: the online form summed them on the VALU + two cross-lane steps): one MFMA per 32 keys
    // against a fragment of ones sums exactly the 16-bit P that multiplies V and arrives reduced over the lanes.  Masked (partial-tile) keys already have
    // P = 0 (their scores were -inf), so the ones fragment needs no masking.
    mma16(lacc, ones, pf[0]);
    mma16(lacc, ones, pf[1]);
    slot = next_slot(slot);
  }
  const float l = lacc[0];
  if constexpr (FIXED) m_run = p.fixed_max / p.scale;                   // the parts path stores m in raw-score units
  if (p.parts > 1) {
    float* W = p.part_ws + ((((long)b * p.H + h) * p.parts + part) * (nw * 16) + wave * 16 + l15) * 66;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) W[dt * 16 + g * 4 + r] = o[dt][r];
    if (g == 0) { W[64] = m_run; W[65] = l; }
    return;
  }
  // a row whose probabilities sum to 0 (every key masked; or every P flushed to zero) or to inf has no softmax: zeros + the range-guard bit instead of 0 / 0 = NaN actions
  const bool l_bad = !(l > 0.f) || vt_nonfinite(l);
  const float inv = l_bad ? 0.f : 1.0f / l;
  if (l_bad && q < p.Nq) vt_range_note(p.range_flag, VT_RANGE_ATTN_EMPTY);
  if (q < p.Nq) {
    uint16_t* O = reinterpret_cast<uint16_t*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 t;
      t.x = kvt_pk<T>(o[dt][0] * inv, o[dt][1] * inv);
      t.y = kvt_pk<T>(o[dt][2] * inv, o[dt][3] * inv);
      *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = t;
    }
  }
}

// merge the key-range parts of a (batch, head): O = sum_p e^{(m_p - m) c} o_p / sum_p e^{(m_p - m) c} l_p, c = scale*log2(e) (exp2 domain, as above)
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_ws, uint16_t* __restrict__ O, long o_bs, long o_rs, int H, int Nq, int rows_pad,
                                                          int parts, float cscale, unsigned* range_flag) {
  // one thread per (row, group of 4 d): 16 rows x 16 groups per block, grid.x blocks of 16 rows (at batch 1 a single block per (b, h) walked
  // 67 x 16 items serially: 38 us per call, 2.6 ms of the 31 ms robot step)
  const int b = blockIdx.z, h = blockIdx.y;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), d4 = (threadIdx.x & 15) * 4;
  if (row >= Nq) return;
  const float* W = part_ws + ((((long)b * H + h) * parts) * rows_pad + row) * 66;
  // every part's (m, l, o[4]) in ONE round of loads (parts <= 16, checked by the launcher): the partials were written by blocks on other XCDs, a
  // dependent second pass over them costs a second fabric round trip (the two-pass form of this loop: 9.2 us per call at batch 1)
  float mp[16], lp[16];
  float2 oa[16], ob[16];
#pragma unroll
  for (int pi = 0; pi < 16; ++pi) {
    mp[pi] = -INFINITY; lp[pi] = 0.f; oa[pi] = make_float2(0.f, 0.f); ob[pi] = make_float2(0.f, 0.f);
    if (pi < parts) {
      const float* Wp = W + (long)pi * rows_pad * 66;
      const float2 ml = *reinterpret_cast<const float2*>(Wp + 64);
      mp[pi] = ml.x; lp[pi] = ml.y;
      oa[pi] = *reinterpret_cast<const float2*>(Wp + d4);
      ob[pi] = *reinterpret_cast<const float2*>(Wp + d4 + 2);
    }
  }
  float m = -INFINITY;
#pragma unroll
  for (int pi = 0; pi < 16; ++pi) m = fmaxf(m, mp[pi]);
  float l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pi = 0; pi < 16; ++pi) {                 // part order: the same sums as the two-pass loop
    const float f = (mp[pi] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mp[pi] - m) * cscale);
    l += f * lp[pi];
    acc[0] += f * oa[pi].x; acc[1] += f * oa[pi].y; acc[2] += f * ob[pi].x; acc[3] += f * ob[pi].y;
  }
  const bool l_bad = !(l > 0.f) || vt_nonfinite(l);
  const float inv = l_bad ? 0.f : 1.0f / l;
  if (l_bad) vt_range_note(range_flag, VT_RANGE_ATTN_EMPTY);
  uint2 t;
  t.x = kvt_pk<T>(acc[0] * inv, acc[1] * inv);
  t.y = kvt_pk<T>(acc[2] * inv, acc[3] * inv);
  *reinterpret_cast<uint2*>(O + (long)b * o_bs + (long)row * o_rs + h * 64 + d4) = t;
}

// small-shape fallbacks: row-major projections [M][ld] (head h at columns h*64..) -> the tile stream.  One block per (tile, h).
// K part: a row copy.
__global__ __launch_bounds__(256) void retile_k_kernel(const bf16_t* __restrict__ Ksrc, long ld, bf16_t* __restrict__ KV, int M, int T) {
  const int h = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const bf16_t* src = Ksrc + (long)h * 64;
  bf16_t* dst = KV + ((long)h * T + t) * 8192;
  for (int e = tid; e < 64 * 8; e += 256) {            // 64 rows x 8 chunks of 8 d
    const int r = e >> 3, c = e & 7, m = t * 64 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < M) v = *reinterpret_cast<const uint4*>(src + (long)m * ld + c * 8);
    *reinterpret_cast<uint4*>(dst + r * 64 + c * 8) = v;
  }
}
// Vt part: transpose through LDS, zero padded for m >= M.
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ V, long ld, bf16_t* __restrict__ KV, int M, int T) {
  __shared__ bf16_t tile[64][66];
  const int h = blockIdx.y, m0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const bf16_t* src = V + (long)h * 64;
  for (int e = tid; e < 64 * 8; e += 256) {            // 64 rows x 8 chunks of 8 d
    const int t = e >> 3, c = e & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m0 + t < M) v = *reinterpret_cast<const uint4*>(src + (long)(m0 + t) * ld + c * 8);
    const bf16_t* ve = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[c * 8 + j][t] = ve[j];
  }
  __syncthreads();
  bf16_t* dst = KV + ((long)h * T + blockIdx.x) * 8192 + 4096;
  for (int e = tid; e < 64 * 32; e += 256) {           // 64 d rows x 32 pairs of keys
    const int d = e >> 5, t2 = (e & 31) * 2;
    const uint32_t v = (uint32_t)tile[d][t2] | ((uint32_t)tile[d][t2 + 1] << 16);
    *reinterpret_cast<uint32_t*>(dst + d * 64 + vt_kpos(t2)) = v;
  }
}
// both halves of the tile stream in ONE launch (blockIdx.z: 0 = K rows, 1 = Vt): the body of the two kernels above
__global__ __launch_bounds__(256) void retile_kv_kernel(const bf16_t* __restrict__ Ksrc, const bf16_t* __restrict__ Vsrc, long ld, bf16_t* __restrict__ KV, int M, int T) {
  __shared__ bf16_t tile[64][66];
  const int h = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  if (blockIdx.z == 0) {
    const bf16_t* src = Ksrc + (long)h * 64;
    bf16_t* dst = KV + ((long)h * T + t) * 8192;
    for (int e = tid; e < 64 * 8; e += 256) {
      const int r = e >> 3, c = e & 7, m = t * 64 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < M) v = *reinterpret_cast<const uint4*>(src + (long)m * ld + c * 8);
      *reinterpret_cast<uint4*>(dst + r * 64 + c * 8) = v;
    }
    return;
  }
  const int m0 = t * 64;
  const bf16_t* src = Vsrc + (long)h * 64;
  for (int e = tid; e < 64 * 8; e += 256) {
    const int r = e >> 3, c = e & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m0 + r < M) v = *reinterpret_cast<const uint4*>(src + (long)(m0 + r) * ld + c * 8);
    const bf16_t* ve = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[c * 8 + j][r] = ve[j];
  }
  __syncthreads();
  bf16_t* dst = KV + ((long)h * T + t) * 8192 + 4096;
  for (int e = tid; e < 64 * 32; e += 256) {
    const int d = e >> 5, t2 = (e & 31) * 2;
    const uint32_t v = (uint32_t)tile[d][t2] | ((uint32_t)tile[d][t2 + 1] << 16);
    *reinterpret_cast<uint32_t*>(dst + d * 64 + vt_kpos(t2)) = v;
  }
}

}  // namespace

static int g_vt_attn_fixed = 1;        // VLATOUCH_ATTN_FIXEDMAX / vt_tune(6, .): fixed-maximum softmax where a score bound is known
// the environment default is read ONCE, before the first explicit setting or launch (an explicit vt_tune(6, .) made before the first launch used to be
// overwritten by the launch's own lazy read of the environment)
static void attn_env_once() {
  static const bool init = [] { const char* e = getenv("VLATOUCH_ATTN_FIXEDMAX"); if (e) g_vt_attn_fixed = atoi(e) != 0; return true; }();
  (void)init;
}
void vt_attn_kvt_tune(int value) { attn_env_once(); g_vt_attn_fixed = value != 0; }
int vt_attn_kvt_fixed_enabled() { attn_env_once(); return g_vt_attn_fixed; }

int vt_attn_kvt_launch(const VtAttnKvtParams& p, hipStream_t s) {
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0 || (long)p.T * 64 < (long)p.B * p.Nk || p.q_rs % 8) return VT_ERR_ARG;
  if (p.dtype != 0 && p.dtype != VT_BF16 && p.dtype != VT_F16) return VT_ERR_UNSUPPORTED;
  const bool f16 = p.dtype == VT_F16;             // 0 (unset) = bf16
  int nw = 4, best = 1 << 30;
  for (int w = 4; w <= 8; ++w) {
    const int rows = w * 16, padded = (p.Nq + rows - 1) / rows * rows;
    if (padded < best) { best = padded; nw = w; }
  }
  const int qblocks = (p.Nq + nw * 16 - 1) / (nw * 16);
  // algorithmic work of one call: every (batch, head) streams its sample's K and Vt tiles once (16 KiB per 64 keys) + Q in, O out
  const double kv_bytes = (double)p.B * p.H * (((long)p.Nk + 63) / 64) * 16384.0, qo_bytes = 2.0 * p.B * p.H * p.Nq * 64 * 2.0;
  VtProfScope prof(4, 4.0 * p.B * p.H * (double)p.Nq * p.Nk * 64, kv_bytes + qo_bytes, s);
  // VLATOUCH_ATTN_RING: 0 = the 2-stage whole-tile kernel, 4 / 5 = half-tile ring with counted waits (default 5)
  static const int ring = [] { const char* e = getenv("VLATOUCH_ATTN_RING"); return e ? atoi(e) : 5; }();
  // fixed-maximum softmax (see attn_kvt_ring_kernel): only with a finite load-time bound small enough that exp(-2B) stays a normal number
  attn_env_once();
  // (IEEE fp16 probabilities: P = 2^15 exp(s - B) in [2^15 e^-2B, 2^15] must stay a NORMAL fp16 (>= 2^-14) for every admissible score, i.e. e^-2B >= 2^-29, B <= 10.05 —
  // with the bound at 16 a row whose scores all sit ~27 under it flushed every P to zero (l = 0 -> NaN) and rows 20 under it lost their tail to subnormals)
  const bool fixed = g_vt_attn_fixed && ring == 5 && p.fixed_max > 0.f && p.fixed_max <= (f16 ? 10.f : 40.f);
  // A/B: extra (unused) dynamic LDS per block lowers the blocks per CU from 4 (4 x 40 KiB = the whole CU) so that a GEMM block of the other in-flight
  // batch can share the CU (VLATOUCH_ATTN_LDS_PAD bytes: 13000 -> 3 blocks, 40000 -> 2 blocks)
  static const int lds_pad = [] { const char* e = getenv("VLATOUCH_ATTN_LDS_PAD"); return e ? atoi(e) : 0; }();
  // nt cache policy (aux = 2) on the K / Vt tile DMA: every tile is read once per launch and the stream (16 GB over the 14 image layers) outlives every
  // cache — 121 -> 104 us per launch averaged over the layers (62 -> 72 % of the HBM roof), full 423 -> 431 chunks/s; VLATOUCH_KVT_NT=0 for A/B
  static const int kv_nt = [] { const char* e = getenv("VLATOUCH_KVT_NT"); return e ? atoi(e) : 1; }();
#define VT_KVT_GO_T(T, grid) \
  do { if (fixed && kv_nt) hipLaunchKernelGGL((attn_kvt_ring_kernel<T, 5, true, 2>), grid, dim3(64 * nw), lds_pad, s, p); \
       else if (fixed) hipLaunchKernelGGL((attn_kvt_ring_kernel<T, 5, true>), grid, dim3(64 * nw), lds_pad, s, p); \
       else if (ring == 5 && kv_nt) hipLaunchKernelGGL((attn_kvt_ring_kernel<T, 5, false, 2>), grid, dim3(64 * nw), lds_pad, s, p); \
       else if (ring == 5) hipLaunchKernelGGL((attn_kvt_ring_kernel<T, 5, false>), grid, dim3(64 * nw), lds_pad, s, p); \
       else if (ring == 4) hipLaunchKernelGGL((attn_kvt_ring_kernel<T, 4, false>), grid, dim3(64 * nw), 0, s, p); \
       else if (ring == 3) hipLaunchKernelGGL((attn_kvt_ring_kernel<T, 3, false>), grid, dim3(64 * nw), 0, s, p); \
       else hipLaunchKernelGGL(attn_kvt_kernel<T>, grid, dim3(64 * nw), 0, s, p); } while (0)
#define VT_KVT_GO(grid) do { if (f16) VT_KVT_GO_T(half_t, grid); else VT_KVT_GO_T(bf16_t, grid); } while (0)
  if (p.parts > 1) {
    if (qblocks != 1 || !p.part_ws || p.parts > 16) return VT_ERR_ARG;
    VT_KVT_GO(dim3(p.parts, p.H, p.B));
    if (f16) hipLaunchKernelGGL(attn_combine_kernel<half_t>, dim3((p.Nq + 15) / 16, p.H, p.B), dim3(256), 0, s, p.part_ws, (uint16_t*)p.O, p.o_bs, p.o_rs, p.H, p.Nq, nw * 16, p.parts,
                                p.scale * 1.4426950408889634f, p.range_flag);
    else hipLaunchKernelGGL(attn_combine_kernel<bf16_t>, dim3((p.Nq + 15) / 16, p.H, p.B), dim3(256), 0, s, p.part_ws, (uint16_t*)p.O, p.o_bs, p.o_rs, p.H, p.Nq, nw * 16, p.parts,
                            p.scale * 1.4426950408889634f, p.range_flag);
    return vt_check_launch();
  }
  dim3 grid(qblocks, p.H, p.B);
  VT_KVT_GO(grid);
#undef VT_KVT_GO_T
#undef VT_KVT_GO
  return vt_check_launch();
}

int vt_k_retile_kv(const void* Ksrc, const void* Vsrc, long ld, void* KV, int M, int T, int H, hipStream_t s) {
  if ((long)T * 64 < M || ld % 8) return VT_ERR_ARG;
  if (Ksrc && Vsrc) hipLaunchKernelGGL(retile_kv_kernel, dim3(T, H, 2), dim3(256), 0, s, (const bf16_t*)Ksrc, (const bf16_t*)Vsrc, ld, (bf16_t*)KV, M, T);
  else if (Ksrc) hipLaunchKernelGGL(retile_k_kernel, dim3(T, H), dim3(256), 0, s, (const bf16_t*)Ksrc, ld, (bf16_t*)KV, M, T);
  else if (Vsrc) hipLaunchKernelGGL(transpose_v_kernel, dim3(T, H), dim3(256), 0, s, (const bf16_t*)Vsrc, ld, (bf16_t*)KV, M, T);
  return vt_check_launch();
}
