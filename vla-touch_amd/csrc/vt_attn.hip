// vt_attn.hip — flash-style attention for head_dim 64 on gfx950 (DINOv2 self-attention, RDT self- and
// cross-attention with an optional key mask).  One workgroup = 4 waves = 64 query rows of one (batch, head);
// K tiles (64 keys) are staged row-major in XOR-swizzled LDS, V tiles are staged TRANSPOSED (Vt[d][key]) so
// both MFMA operands are read as contiguous 8-element fragments.  Per wave (16 query rows):
//   S^T tile = mfma(K_frag, Q_frag)  -> lane holds S[q = lane&15][key = kt*16 + (lane>>4)*4 + r]
//   O^T tile = mfma(Vt_frag, P_frag) -> lane holds O[q = lane&15][d  = dt*16 + (lane>>4)*4 + r]
// The k-index -> key assignment inside a 32-key block is (j>>2)*16 + g*4 + (j&3), which makes the P fragment
// exactly the lane's own S registers (no cross-lane movement); Vt is read with the same assignment.
// Online softmax state (m, l) lives per q row, replicated over the 4 lanes sharing lane&15.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

template <typename T> struct QLoad;
template <> struct QLoad<bf16_t> {
  __device__ static __forceinline__ void ld(Frag<bf16_t>& f, const bf16_t* p, bool ok) {
    f.v = ok ? *reinterpret_cast<const short8_t*>(p) : (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
  }
};
template <> struct QLoad<half_t> {
  __device__ static __forceinline__ void ld(Frag<half_t>& f, const half_t* p, bool ok) {
    f.v = ok ? *reinterpret_cast<const short8_t*>(p) : (short8_t){0, 0, 0, 0, 0, 0, 0, 0};
  }
};
template <> struct QLoad<float> {
  __device__ static __forceinline__ void ld(Frag<float>& f, const float* p, bool ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = ok ? p[j] : 0.f;
  }
};

template <typename T> struct PackP;
template <> struct PackP<bf16_t> {
  __device__ static __forceinline__ void pack(Frag<bf16_t>& f, const float p[8]) {
    const uint4 w = make_uint4(pk_bf16(p[0], p[1]), pk_bf16(p[2], p[3]), pk_bf16(p[4], p[5]), pk_bf16(p[6], p[7]));
    f.v = __builtin_bit_cast(short8_t, w);
  }
};
template <> struct PackP<half_t> {
  __device__ static __forceinline__ void pack(Frag<half_t>& f, const float p[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (short)f2h(p[j]).b;
  }
};
template <> struct PackP<float> {
  __device__ static __forceinline__ void pack(Frag<float>& f, const float p[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = p[j];
  }
};

template <typename T> struct VtRead;   // 4 consecutive keys of one d row
template <> struct VtRead<bf16_t> {
  __device__ static __forceinline__ void rd(Frag<bf16_t>& f, int half, const char* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    f.v[half * 4 + 0] = (short)(t.x & 0xffff); f.v[half * 4 + 1] = (short)(t.x >> 16);
    f.v[half * 4 + 2] = (short)(t.y & 0xffff); f.v[half * 4 + 3] = (short)(t.y >> 16);
  }
};
template <> struct VtRead<half_t> {
  __device__ static __forceinline__ void rd(Frag<half_t>& f, int half, const char* p) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    f.v[half * 4 + 0] = (short)(t.x & 0xffff); f.v[half * 4 + 1] = (short)(t.x >> 16);
    f.v[half * 4 + 2] = (short)(t.y & 0xffff); f.v[half * 4 + 3] = (short)(t.y >> 16);
  }
};
template <> struct VtRead<float> {
  __device__ static __forceinline__ void rd(Frag<float>& f, int half, const char* p) {
    const float4_t t = *reinterpret_cast<const float4_t*>(p);
    f.v[half * 4 + 0] = t[0]; f.v[half * 4 + 1] = t[1]; f.v[half * 4 + 2] = t[2]; f.v[half * 4 + 3] = t[3];
  }
};

// blockDim.x = 64 * NW (NW = 4..8 waves); a block owns 16*NW query rows of one (batch, head); NW is chosen by the host
// to minimise padded query rows (e.g. 5 waves = 80 rows for RDT's 67 queries, so its 4 374-key cross-attention streams
// K/V once per (batch, head) instead of twice).
// HD = head dimension, 64 or 96 (SigLIP's 72-wide heads run zero-padded to 96: its QKV / projection weights are packed that way).
template <typename T, int HD>
__global__ __launch_bounds__(512) void attn_kernel(const VtAttnParams p) {
  constexpr int KT = 64;
  constexpr int ES = sizeof(T);
  constexpr int EPC = Elem<T>::EPC;
  constexpr int EPR = 128 / ES;            // elements per 128-B LDS row (64 bf16 / 32 f32)
  constexpr int SUB = (HD + EPR - 1) / EPR; // 128-B sub-tiles per key row
  constexpr int CPK = HD / EPC;            // 16-B chunks per key row (8 / 16)
  constexpr int VSTR = 68 * ES;            // Vt row stride in bytes (64 keys + 4 pad)
  __shared__ __attribute__((aligned(16))) char smem[SUB * KT * 128 + HD * VSTR];
  char* Ks = smem;
  char* Vt = smem + SUB * KT * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int nthreads = blockDim.x;
  const int q = blockIdx.x * (nthreads >> 2) + wave * 16 + l15;
  const T* Q = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_bs + (long)h * p.q_hs;
  const T* K = reinterpret_cast<const T*>(p.K) + (long)b * p.k_bs + (long)h * p.k_hs;
  const T* V = reinterpret_cast<const T*>(p.V) + (long)b * p.v_bs + (long)h * p.v_hs;
  const uint8_t* km = p.kmask ? p.kmask + (long)b * p.km_bs : nullptr;

  constexpr int NKS = HD / 32, NDT = HD / 16;      // 32-deep k-steps of Q.K, 16-row tiles of the output's d dimension
  Frag<T> qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) QLoad<T>::ld(qf[ks], Q + (long)q * p.q_rs + ks * 32 + g * 8, q < p.Nq);

  float4_t o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float cscale = p.scale * 1.4426950408889634f;

  const int ntiles = (p.Nk + KT - 1) / KT;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int key0 = tile * KT;
    __syncthreads();   // previous tile fully consumed
    // ---- stage K (row-major, swizzled) and V (transposed)
    for (int ci = tid; ci < KT * CPK; ci += nthreads) {
      const int key = ci / CPK, cidx = ci - key * CPK;
      const bool ok = key0 + key < p.Nk;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (ok) {
        kv = *reinterpret_cast<const uint4*>(K + (long)(key0 + key) * p.k_rs + cidx * EPC);
        vv = *reinterpret_cast<const uint4*>(V + (long)(key0 + key) * p.v_rs + cidx * EPC);
      }
      const int sub = cidx >> 3, cc = cidx & 7;
      *reinterpret_cast<uint4*>(Ks + sub * (KT * 128) + key * 128 + swz(key, cc) * 16) = kv;
      const T* ve = reinterpret_cast<const T*>(&vv);
#pragma unroll
      for (int j = 0; j < EPC; ++j) *reinterpret_cast<T*>(Vt + (cidx * EPC + j) * VSTR + key * ES) = ve[j];
    }
    __syncthreads();

    // ---- S = K Q^T  (4 key sub-tiles of 16)
    float4_t sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int d0 = ks * 32 + g * 8;
        Frag<T> kf;
        lds_frag(kf, Ks + (d0 / EPR) * (KT * 128), kt * 16 + l15, (d0 % EPR) / 8);
        mma16(sacc[kt], kf, qf[ks]);
      }
    }
    // ---- online softmax in the exp2 domain: p = exp2(s*c - m*c), c = scale*log2(e) > 0 (the max is taken on raw scores);
    // masking (partial last tile, key mask) is a block-uniform slow path
    float sv[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[kt * 4 + r] = sacc[kt][r];
    if (km || key0 + KT > p.Nk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kidx = key0 + kt * 16 + g * 4 + r;
          bool ok = kidx < p.Nk;
          if (ok && km) ok = km[kidx] != 0;
          if (!ok) sv[kt * 4 + r] = -INFINITY;
        }
    }
    float mx = sv[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new != m_run)) {          // rescale only when some row's running max moved
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * cscale);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      m_run = m_new;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : m_run * cscale;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -mc)); psum += sv[i]; }
    l_run += psum;
    // ---- O += P V   (two 32-key blocks)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float pj[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pj[j] = sv[(kb * 2 + (j >> 2)) * 4 + (j & 3)];
      Frag<T> pf;
      PackP<T>::pack(pf, pj);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Frag<T> vf;
        const char* row = Vt + (dt * 16 + l15) * VSTR;
        VtRead<T>::rd(vf, 0, row + (kb * 32 + g * 4) * ES);
        VtRead<T>::rd(vf, 1, row + (kb * 32 + 16 + g * 4) * ES);
        mma16(o[dt], vf, pf);
      }
    }
  }
  // ---- finalise
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (q < p.Nq) {
    T* O = reinterpret_cast<T*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[dt * 16 + g * 4 + r] = Elem<T>::from_f(o[dt][r] * inv);
  }
}

// ---- 16-bit self-attention with DMA-staged, double-buffered K / V tiles (DINOv2 257 tokens, SigLIP 729 tokens x 72-wide heads padded to 96, RDT
// self-attention).  Same arithmetic and fragment conventions as attn_kernel above; what changes is how the tiles reach the MFMAs:
//   * K and V tiles (64 keys) go HBM/L2 -> LDS by DMA (16 B per lane, 8 rows x 128 B per wave instruction, XOR swizzle on the SOURCE address),
//     BOTH row-major [key][d] — attn_kernel transposed V through registers with eight 2-byte LDS stores per 16-byte chunk, which alone cost
//     more LDS instructions than everything else in the tile;
//   * the Vt fragment (4 consecutive keys of one d column) is read from the row-major V tile with ds_read_b64_tr_b16: the 16 lanes of a lane
//     group address a [4 keys][16 d] block (lane i: key i/4, d columns (i%4)*4 .. +3) and the hardware hands lane i the 4 keys of column i;
//   * a ring of three LDS stages: two tiles are in flight behind the one being consumed, waits are counted (vmcnt), ONE raw barrier per tile.
// No key mask here (masked calls keep attn_kernel).  Keys past Nk in the last tile are clamped loads, masked in the softmax.
typedef __attribute__((address_space(3))) void lds_void_a;
typedef const __attribute__((address_space(1))) void glb_void_a;
typedef __attribute__((ext_vector_type(4))) short short4_t;

// 16-deep tail step of a head dimension that is not a multiple of 32 (SigLIP's 72-wide heads run padded to 80 = 2 x 32 + 16): v_mfma_f32_16x16x16,
// a lane (row l15, group g) holds the 4 elements k = g*4 .. g*4+3 of both operands
typedef __attribute__((ext_vector_type(4))) _Float16 half4v_t;
template <typename T> __device__ __forceinline__ void mma16_k16(float4_t& acc, const short4_t a, const short4_t b);
template <> __device__ __forceinline__ void mma16_k16<bf16_t>(float4_t& acc, const short4_t a, const short4_t b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16_k16<half_t>(float4_t& acc, const short4_t a, const short4_t b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4v_t, a), __builtin_bit_cast(half4v_t, b), acc, 0, 0, 0);
}

#ifdef VLATOUCH_BENCH_BUILD      // timing-only ablations of attn16_kernel (tools/attn_abl.sh; garbage results): 1 = no softmax arithmetic, 2 = no P V, 4 = no Q K^T, 8 = no K / V staging after the first tiles
__device__ int d_attn_abl = 0;
__device__ long long* d_attn_tbuf = nullptr;   // optional phase time stamps of attn16u_kernel (tools/attn_phases.py): 4 x s_memrealtime per block
#define VT_ATTN_ABL(bit) (d_attn_abl & (bit))
#else
#define VT_ATTN_ABL(bit) 0
#endif

template <typename T, int HD>
__global__ __launch_bounds__(512) void attn16_kernel(const VtAttnParams p) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int KT = 64;
  constexpr int NKS = HD / 32, NDT = HD / 16;
  constexpr bool TAIL16 = (HD % 32) == 16;            // one 16-deep step behind the NKS 32-deep ones
  static_assert(HD >= 64 && HD <= 128 && HD % 16 == 0 && (HD % 32 == 0 || HD % 32 == 16), "head dimension: 64 .. 128, multiple of 16");
  // LDS image of a K (or V) tile: MAIN = [64 keys][128 B] for d 0..63 (16-byte chunks XOR-swizzled by the row) followed by a compact TAIL =
  // [64 keys][TW bytes] for d 64.. (linear: its reads are contiguous as they are) — 8 / 10 / 12 KiB for 64 / 80 / 96-wide heads, so THREE stages
  // (K + V each) still leave two blocks per CU
  constexpr int TW = (HD - 64) * 2;                   // tail bytes per key row: 0, 32, 64
  constexpr int MAINB = KT * 128;
  constexpr int TILE = MAINB + KT * TW;
  constexpr int STAGE = 2 * TILE;                     // K then V
  constexpr int NST = 3;
  constexpr int TP = KT * TW / 1024;                  // tail DMA pieces per tile (a piece = 1 KiB = 1024 / TW rows)
  constexpr int PT = 8 + TP;                          // pieces per tile
  constexpr int PIECES = 2 * PT;                      // per stage
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * (nw * 16) + wave * 16 + l15;
  const T* Q = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_bs + (long)h * p.q_hs;
  const T* K = reinterpret_cast<const T*>(p.K) + (long)b * p.k_bs + (long)h * p.k_hs;
  const T* V = reinterpret_cast<const T*>(p.V) + (long)b * p.v_bs + (long)h * p.v_hs;

  Frag<T> qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) QLoad<T>::ld(qf[ks], Q + (long)q * p.q_rs + ks * 32 + g * 8, q < p.Nq);
  short4_t q16 = {0, 0, 0, 0};
  if constexpr (TAIL16) { if (q < p.Nq) q16 = *reinterpret_cast<const short4_t*>(Q + (long)q * p.q_rs + NKS * 32 + g * 4); }
  // retire the Q loads before the first DMA (an ordinary load pending beside LDS-DMA makes hipcc drain the whole queue at its first use)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks].v));
  asm volatile("" : "+v"(q16));

  // DMA plan: the PIECES 1-KiB pieces of a stage are dealt round-robin over the waves (the counted wait uses the per-wave count).  Main
  // pieces: 8 rows x 128 B, lane -> (row, chunk position), it fetches the chunk whose swizzled position is its own; tail pieces: 1024 / TW
  // rows x TW bytes, linear.
  const int my_pieces = (PIECES - wave + nw - 1) / nw;                 // pieces i = wave, wave + nw, ...
  auto stage = [&](const int slot, const int tile) {
    const int key0 = tile * KT;
    for (int i = wave; i < PIECES; i += nw) {
      const bool isv = i >= PT;
      const int j = isv ? i - PT : i;
      const T* base = isv ? V : K;
      const long rs = isv ? p.v_rs : p.k_rs;
      char* dst = smem + slot * STAGE + (isv ? TILE : 0);
      if (TP == 0 || j < 8) {
        const int r = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_global_load_lds((glb_void_a*)(base + (long)min(key0 + r, p.Nk - 1) * rs + c * 8), (lds_void_a*)(dst + j * 1024), 16, 0, 0);
      } else if constexpr (TP > 0) {
        constexpr int CPR = TW / 16;                  // 16-byte chunks per tail row: 2 or 4
        const int r = (j - 8) * (64 / CPR) + lane / CPR;
        const int c = lane % CPR;
        __builtin_amdgcn_global_load_lds((glb_void_a*)(base + (long)min(key0 + r, p.Nk - 1) * rs + 64 + c * 8), (lds_void_a*)(dst + MAINB + (j - 8) * 1024), 16, 0, 0);
      }
    }
  };
  auto wait_own = [&](const bool younger_in_flight) {   // this wave's pieces of the oldest outstanding stage have landed
    if (!younger_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (my_pieces == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (my_pieces == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (my_pieces == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (my_pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (my_pieces == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (my_pieces == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  float4_t o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float cscale = p.scale * 1.4426950408889634f;
  const int ntiles = (p.Nk + KT - 1) / KT;

  // ring of NST = 3 stages, two tiles in flight behind the one being consumed; ONE barrier per tile: it publishes tile t (every wave waited
  // for its own pieces) and proves that everybody is done with tile t-1, whose slot the DMA of tile t+2 then takes
  stage(0, 0);
  if (ntiles > 1) stage(1, 1);
  int slot = 0;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int key0 = tile * KT;
    wait_own(tile + 1 < ntiles);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (tile + 2 < ntiles && !VT_ATTN_ABL(8)) stage(slot == 0 ? 2 : slot - 1, tile + 2);
    const char* Ks = smem + slot * STAGE;
    const char* Vs = Ks + TILE;
    slot = slot == NST - 1 ? 0 : slot + 1;

    float4_t sacc[4], tacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
      tacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (VT_ATTN_ABL(4)) continue;
      const int row = kt * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        Frag<T> kf;
        if (ks < 2) lds_frag(kf, Ks, row, ks * 4 + g);                                                  // d 0..63: the swizzled main image
        else kf.v = *reinterpret_cast<const short8_t*>(Ks + MAINB + row * TW + (ks - 2) * 64 + g * 16);  // d 64..: the linear tail
        mma16(sacc[kt], kf, qf[ks]);
      }
      if constexpr (TAIL16) {
        // into its OWN accumulator, added on the VALU below: chained straight behind the 8-pass 16x16x32 MFMAs on the same accumulator, the
        // 4-pass 16x16x16 read registers 0..1 of the last key tile before they were written (scores of keys 48 + 4g + {0, 1} lost their d < 64
        // part; tools/_dbg pinpointed it) — hipcc 7.2 does not pad that SrcC hazard between the two instruction lengths
        const short4_t k16 = *reinterpret_cast<const short4_t*>(Ks + MAINB + row * TW + (NKS - 2) * 64 + g * 8);
        tacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        mma16_k16<T>(tacc[kt], k16, q16);
      }
    }
    float sv[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[kt * 4 + r] = TAIL16 ? sacc[kt][r] + tacc[kt][r] : sacc[kt][r];
    if (key0 + KT > p.Nk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + kt * 16 + g * 4 + r >= p.Nk) sv[kt * 4 + r] = -INFINITY;
    }
    if (!VT_ATTN_ABL(1)) {
    float mx = sv[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (__any(m_new != m_run)) {
      const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * cscale);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      m_run = m_new;
    }
    const float mc = (m_run == -INFINITY) ? 0.f : m_run * cscale;
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -mc)); psum += sv[i]; }
    l_run += psum;
    } else { l_run = 1.f; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (VT_ATTN_ABL(2)) break;
      float pj[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pj[j] = sv[(kb * 2 + (j >> 2)) * 4 + (j & 3)];
      Frag<T> pf;
      PackP<T>::pack(pf, pj);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        // lane i of group g addresses key (kb*32 [+16] + g*4 + i/4), d columns dt*16 + (i%4)*4 .. +3 of the row-major V tile and receives
        // the 4 keys kb*32 [+16] + g*4 .. +3 of column dt*16 + i
        const int dcol = dt * 16 + (l15 & 3) * 4;
        Frag<T> vf;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int key = kb * 32 + hh * 16 + g * 4 + (l15 >> 2);
          const char* a = dt < 4 ? Vs + key * 128 + (((dcol >> 3) ^ ((key >> 1) & 7)) * 16) + (dcol & 7) * 2
                                 : Vs + MAINB + key * TW + (dcol - 64) * 2;
          const short4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a);
          vf.v[hh * 4 + 0] = t[0]; vf.v[hh * 4 + 1] = t[1]; vf.v[hh * 4 + 2] = t[2]; vf.v[hh * 4 + 3] = t[3];
        }
        mma16(o[dt], vf, pf);
      }
    }
  }
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (q < p.Nq) {
    T* O = reinterpret_cast<T*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      T ov[4] = {Elem<T>::from_f(o[dt][0] * inv), Elem<T>::from_f(o[dt][1] * inv), Elem<T>::from_f(o[dt][2] * inv), Elem<T>::from_f(o[dt][3] * inv)};
      *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = *reinterpret_cast<const uint2*>(ov);
    }
  }
}


// ---- attn16u_kernel: attn16_kernel with the tile loop unrolled over the three ring slots.  With the slot a compile-time constant every LDS
// address of the tile is (per-lane offset computed ONCE before the loop) + (immediate): the 60 address instructions per tile of attn16_kernel
// (a third of its VALU issue slots; the kernel is VALU-issue-bound in its arithmetic, tools/attn_abl.sh) disappear.  The DMA source addresses
// are (uniform tile base, SALU) + (per-lane piece offset, computed once) except in the last, clamped tile, and the first two stages are issued
// BEFORE the Q fragments are loaded so that a block pays one memory round trip before its first MFMA, not two.

// max / sum over the four 16-lane rows of a wave (lanes that differ in bits 4 and 5), result in every lane, on v_permlane16_swap / v_permlane32_swap
// (VALU) instead of two __shfl_xor = ds_bpermute round trips (round 5: the same change took the cached cross-attention's online softmax from 114 to 104 us)
template <bool MAX> __device__ __forceinline__ float rows4_reduce(float v) {
#ifdef VLATOUCH_ATTN_SHFL
  const float a = __shfl_xor(v, 16, 64);
  v = MAX ? fmaxf(v, a) : v + a;
  const float b = __shfl_xor(v, 32, 64);
  return MAX ? fmaxf(v, b) : v + b;
#else
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float a0 = __builtin_bit_cast(float, (unsigned)a[0]), a1 = __builtin_bit_cast(float, (unsigned)a[1]);
  const float w = MAX ? fmaxf(a0, a1) : a0 + a1;
  const unsigned x = __builtin_bit_cast(unsigned, w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const float b0 = __builtin_bit_cast(float, (unsigned)b[0]), b1 = __builtin_bit_cast(float, (unsigned)b[1]);
  return MAX ? fmaxf(b0, b1) : b0 + b1;
#endif
}
template <int I> struct IC { static constexpr int value = I; };

template <typename T, int HD>
__global__ __launch_bounds__(512) void attn16u_kernel(const VtAttnParams p) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int KT = 64;
  constexpr int NKS = HD / 32, NDT = HD / 16;
  constexpr bool TAIL16 = (HD % 32) == 16;
  constexpr int TW = (HD - 64) * 2;
  constexpr int MAINB = KT * 128;
  constexpr int TILE = MAINB + KT * TW;
  constexpr int STAGE = 2 * TILE;
  constexpr int NST = 3;
  constexpr int TP = KT * TW / 1024;
  constexpr int PT = 8 + TP;
  constexpr int PIECES = 2 * PT;
  constexpr int MAXP = (PIECES + 3) / 4;              // pieces per wave at the smallest block (4 waves)
  constexpr int CPR = TW > 0 ? TW / 16 : 1;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * (nw * 16) + wave * 16 + l15;
  const T* Q = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_bs + (long)h * p.q_hs;
  const T* K = reinterpret_cast<const T*>(p.K) + (long)b * p.k_bs + (long)h * p.k_hs;
  const T* V = reinterpret_cast<const T*>(p.V) + (long)b * p.v_bs + (long)h * p.v_hs;
  const int ntiles = (p.Nk + KT - 1) / KT;
#ifdef VLATOUCH_BENCH_BUILD
  const int lin_blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  long long* tb = (d_attn_tbuf && lin_blk < 4096 && tid == 0) ? d_attn_tbuf + (long)lin_blk * 4 : nullptr;
  if (tb) tb[0] = wall_clock64();
#endif

  // DMA plan (as attn16_kernel): the PIECES 1-KiB pieces of a stage are dealt round-robin over the waves.  poff[n] = this lane's element offset
  // inside the K (or V) rows of a tile for the wave's n-th piece.
  const int my_pieces = (PIECES - wave + nw - 1) / nw;
  unsigned poff[MAXP];
#pragma unroll
  for (int n = 0; n < MAXP; ++n) {
    const int i = wave + n * nw;
    const bool isv = i >= PT;
    const int j = isv ? i - PT : i;
    const unsigned rs = (unsigned)(isv ? p.v_rs : p.k_rs);
    if (TP == 0 || j < 8) {
      const int r = j * 8 + (lane >> 3);
      poff[n] = (unsigned)r * rs + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 8);
    } else {
      const int r = (j - 8) * (64 / CPR) + lane / CPR;
      poff[n] = (unsigned)r * rs + 64u + (unsigned)((lane % CPR) * 8);
    }
  }
  auto piece_dst = [&](const int slot, const int i) __attribute__((always_inline)) -> char* {
    const bool isv = i >= PT;
    const int j = isv ? i - PT : i;
    return smem + slot * STAGE + (isv ? TILE : 0) + ((TP == 0 || j < 8) ? j * 1024 : MAINB + (j - 8) * 1024);
  };
  auto stage = [&](const int slot, const int tile) __attribute__((always_inline)) {
    const int key0 = tile * KT;
    if (key0 + KT <= p.Nk) {                          // whole tile: uniform base + per-lane piece offset
      const T* kb = K + (long)key0 * p.k_rs;
      const T* vb = V + (long)key0 * p.v_rs;
#pragma unroll
      for (int n = 0; n < MAXP; ++n) {
        const int i = wave + n * nw;
        if (i < PIECES)
          __builtin_amdgcn_global_load_lds((glb_void_a*)((i >= PT ? vb : kb) + poff[n]), (lds_void_a*)piece_dst(slot, i), 16, 0, 0);
      }
    } else {                                          // last tile: rows clamped to the last key (masked in the softmax)
      for (int i = wave; i < PIECES; i += nw) {
        const bool isv = i >= PT;
        const int j = isv ? i - PT : i;
        const T* base = isv ? V : K;
        const long rs = isv ? p.v_rs : p.k_rs;
        if (TP == 0 || j < 8) {
          const int r = j * 8 + (lane >> 3);
          const int c = (lane & 7) ^ ((r >> 1) & 7);
          __builtin_amdgcn_global_load_lds((glb_void_a*)(base + (long)min(key0 + r, p.Nk - 1) * rs + c * 8), (lds_void_a*)piece_dst(slot, i), 16, 0, 0);
        } else {
          const int r = (j - 8) * (64 / CPR) + lane / CPR;
          const int c = lane % CPR;
          __builtin_amdgcn_global_load_lds((glb_void_a*)(base + (long)min(key0 + r, p.Nk - 1) * rs + 64 + c * 8), (lds_void_a*)piece_dst(slot, i), 16, 0, 0);
        }
      }
    }
  };
  auto wait_own = [&](const bool younger_in_flight) __attribute__((always_inline)) {
    if (!younger_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (my_pieces == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (my_pieces == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (my_pieces == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (my_pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (my_pieces == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (my_pieces == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // first two stages, THEN the Q fragments; everything retired together (an ordinary load pending beside LDS-DMA makes hipcc drain the whole
  // queue at its first use anyway)
  stage(0, 0);
  if (ntiles > 1) stage(1, 1);
  Frag<T> qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) QLoad<T>::ld(qf[ks], Q + (long)q * p.q_rs + ks * 32 + g * 8, q < p.Nq);
  short4_t q16 = {0, 0, 0, 0};
  if constexpr (TAIL16) { if (q < p.Nq) q16 = *reinterpret_cast<const short4_t*>(Q + (long)q * p.q_rs + NKS * 32 + g * 4); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks].v));
  asm volatile("" : "+v"(q16));
#ifdef VLATOUCH_BENCH_BUILD
  if (tb) tb[1] = wall_clock64();
#endif

  // per-lane LDS offsets inside a tile (see attn16_kernel for the fragment conventions)
  //   K fragment (key row kt*16 + l15, 16-byte chunk ks*4 + g, swizzled by (row >> 1) & 7 = (l15 >> 1) & 7):  kt*2048 + koff[ks]
  //   V fragment (key kb*32 + hh*16 + vkey, d columns dt*16 + (l15 & 3)*4 ..):  (kb*32 + hh*16)*128 + voff[dt]   (dt < 4)
  const int ksw = (l15 >> 1) & 7;
  unsigned koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = (unsigned)(l15 * 128 + (((ks * 4 + g) ^ ksw) * 16));
  const unsigned ktail = (unsigned)(MAINB + l15 * TW + g * 16);          // + kt*16*TW + (ks-2)*64; the 16-deep tail step reads ... + g*8 instead
  const unsigned ktail16 = (unsigned)(MAINB + l15 * TW + (NKS - 2) * 64 + g * 8);
  const int vkey = g * 4 + (l15 >> 2);
  const int vsw = ((vkey >> 1) & 7) ^ ((l15 & 3) >> 1);
  unsigned voff[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) voff[dt] = (unsigned)(vkey * 128 + (((dt * 2) ^ vsw) * 16) + (l15 & 1) * 8);
  const unsigned vtail = (unsigned)(MAINB + vkey * TW + (l15 & 3) * 8);  // + (kb*32 + hh*16)*TW + (dt-4)*32
#pragma unroll
  for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(koff[i]));
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(voff[i]));

  float4_t o[NDT];
#pragma unroll
  for (int i = 0; i < NDT; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float cscale = p.scale * 1.4426950408889634f;

  auto body = [&](auto slot_c, const int tile) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr int NEXT = SLOT == 0 ? 2 : SLOT - 1;      // the slot tile t-1 used: free after this tile's barrier
    const int key0 = tile * KT;
    wait_own(tile + 1 < ntiles);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (tile + 2 < ntiles && !VT_ATTN_ABL(8)) stage(NEXT, tile + 2);
    const char* Ks = smem + SLOT * STAGE;
    const char* Vs = Ks + TILE;

    float4_t sacc[4], tacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
      tacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (VT_ATTN_ABL(4)) continue;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        Frag<T> kf;
        if (ks < 2) kf.v = *reinterpret_cast<const short8_t*>(Ks + kt * 2048 + koff[ks]);
        else kf.v = *reinterpret_cast<const short8_t*>(Ks + kt * 16 * TW + (ks - 2) * 64 + ktail);
        mma16(sacc[kt], kf, qf[ks]);
      }
      if constexpr (TAIL16) {
        // own accumulator, added on the VALU below (the SrcC hazard between the 8-pass and 4-pass MFMAs, see attn16_kernel)
        const short4_t k16 = *reinterpret_cast<const short4_t*>(Ks + kt * 16 * TW + ktail16);
        mma16_k16<T>(tacc[kt], k16, q16);
      }
    }
    float sv[16];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[kt * 4 + r] = TAIL16 ? sacc[kt][r] + tacc[kt][r] : sacc[kt][r];
    if (key0 + KT > p.Nk) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (key0 + kt * 16 + g * 4 + r >= p.Nk) sv[kt * 4 + r] = -INFINITY;
    }
    if (!VT_ATTN_ABL(1)) {
      float mx = sv[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
      mx = rows4_reduce<true>(mx);
      const float m_new = fmaxf(m_run, mx);
      if (__any(m_new != m_run)) {
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * cscale);
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
        m_run = m_new;
      }
      const float mc = (m_run == -INFINITY) ? 0.f : m_run * cscale;
      float psum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -mc)); psum += sv[i]; }
      l_run += psum;
    } else { l_run = 1.f; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (VT_ATTN_ABL(2)) break;
      float pj[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pj[j] = sv[(kb * 2 + (j >> 2)) * 4 + (j & 3)];
      Frag<T> pf;
      PackP<T>::pack(pf, pj);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Frag<T> vf;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const char* a = dt < 4 ? Vs + (kb * 32 + hh * 16) * 128 + voff[dt < 4 ? dt : 0]
                                 : Vs + (kb * 32 + hh * 16) * TW + (dt - 4) * 32 + vtail;
          const short4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a);
          vf.v[hh * 4 + 0] = t[0]; vf.v[hh * 4 + 1] = t[1]; vf.v[hh * 4 + 2] = t[2]; vf.v[hh * 4 + 3] = t[3];
        }
        mma16(o[dt], vf, pf);
      }
    }
  };

  for (int tile = 0; tile < ntiles; tile += 3) {
    body(IC<0>{}, tile);
    if (tile + 1 >= ntiles) break;
    body(IC<1>{}, tile + 1);
    if (tile + 2 >= ntiles) break;
    body(IC<2>{}, tile + 2);
  }

#ifdef VLATOUCH_BENCH_BUILD
  if (tb) { asm volatile("" : "+v"(o[0])); tb[2] = wall_clock64(); }
#endif
  float l = l_run;
  l = rows4_reduce<false>(l);
  const float inv = 1.0f / l;
  if (q < p.Nq) {
    T* O = reinterpret_cast<T*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * HD;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      T ov[4] = {Elem<T>::from_f(o[dt][0] * inv), Elem<T>::from_f(o[dt][1] * inv), Elem<T>::from_f(o[dt][2] * inv), Elem<T>::from_f(o[dt][3] * inv)};
      *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = *reinterpret_cast<const uint2*>(ov);
    }
  }
#ifdef VLATOUCH_BENCH_BUILD
  if (tb) tb[3] = wall_clock64();
#endif
}


// ---- attn16g_kernel (round 5): the ViT self-attention loop nest turned inside out.  attn16u_kernel gives a block 16 query rows per wave and walks the
// whole key sequence for them: a SigLIP (image, head) is 6 blocks, each staging the same 233 KB of K / V (tools/attn_abl.sh: a third of the launch is
// re-staging, another third the per-block fixed cost — dispatch, Q loads, first DMA round trip, 12 barriers, O stores).  Here a wave owns G query
// groups of 16 rows (G sets of Q fragments, running (m, l) and O accumulators in registers) and a block walks the key tiles ONCE for all of its
// 16 * nw * G rows: one block per (image, head) for SigLIP (8 waves x 6 groups = 768 >= 729 rows) and DINOv2 @224 (6 x 3 = 288 >= 257).  Inside a
// key tile the G groups are independent instruction streams of one wave: the scheduler runs the Q K^T MFMAs of group i + 1 under the softmax VALU
// work of group i.  Same tile image, DMA plan, fragment conventions and arithmetic (per row: identical operations in identical order) as
// attn16u_kernel, so results are bit-identical to it.  Group gi of wave w = rows (gi * nw + w) * 16 .. + 15 of the block; groups that start past Nq
// are skipped (wave-uniform branch).  The ring slot is a run-time value here (8 address adds per tile, amortised over G groups).
template <int N, typename F> __device__ __forceinline__ void ag_for(F&& f) {
  if constexpr (N > 0) { ag_for<N - 1>(f); f(IC<N - 1>{}); }
}

template <typename T, int HD, int G>
__global__ __launch_bounds__(512) void attn16g_kernel(const VtAttnParams p) {
  static_assert(sizeof(T) == 2, "16-bit types only");
  constexpr int KT = 64;
  constexpr int NKS = HD / 32, NDT = HD / 16;
  constexpr bool TAIL16 = (HD % 32) == 16;
  constexpr int TW = (HD - 64) * 2;
  constexpr int MAINB = KT * 128;
  constexpr int TILE = MAINB + KT * TW;
  constexpr int STAGE = 2 * TILE;
  constexpr int NST = 3;
  constexpr int TP = KT * TW / 1024;
  constexpr int PT = 8 + TP;
  constexpr int PIECES = 2 * PT;
  constexpr int MAXP = (PIECES + 3) / 4;
  constexpr int CPR = TW > 0 ? TW / 16 : 1;
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int row_base = blockIdx.x * (nw * 16 * G) + wave * 16;         // + gi * nw * 16 = first row of this wave's group gi
  const T* Q = reinterpret_cast<const T*>(p.Q) + (long)b * p.q_bs + (long)h * p.q_hs;
  const T* K = reinterpret_cast<const T*>(p.K) + (long)b * p.k_bs + (long)h * p.k_hs;
  const T* V = reinterpret_cast<const T*>(p.V) + (long)b * p.v_bs + (long)h * p.v_hs;
  const int ntiles = (p.Nk + KT - 1) / KT;

  const int my_pieces = (PIECES - wave + nw - 1) / nw;
  unsigned poff[MAXP];
#pragma unroll
  for (int n = 0; n < MAXP; ++n) {
    const int i = wave + n * nw;
    const bool isv = i >= PT;
    const int j = isv ? i - PT : i;
    const unsigned rs = (unsigned)(isv ? p.v_rs : p.k_rs);
    if (TP == 0 || j < 8) {
      const int r = j * 8 + (lane >> 3);
      poff[n] = (unsigned)r * rs + (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 8);
    } else {
      const int r = (j - 8) * (64 / CPR) + lane / CPR;
      poff[n] = (unsigned)r * rs + 64u + (unsigned)((lane % CPR) * 8);
    }
  }
  auto piece_dst = [&](const int slot, const int i) __attribute__((always_inline)) -> char* {
    const bool isv = i >= PT;
    const int j = isv ? i - PT : i;
    return smem + slot * STAGE + (isv ? TILE : 0) + ((TP == 0 || j < 8) ? j * 1024 : MAINB + (j - 8) * 1024);
  };
  auto stage = [&](const int slot, const int tile) __attribute__((always_inline)) {
    const int key0 = tile * KT;
    if (key0 + KT <= p.Nk) {
      const T* kb = K + (long)key0 * p.k_rs;
      const T* vb = V + (long)key0 * p.v_rs;
#pragma unroll
      for (int n = 0; n < MAXP; ++n) {
        const int i = wave + n * nw;
        if (i < PIECES)
          __builtin_amdgcn_global_load_lds((glb_void_a*)((i >= PT ? vb : kb) + poff[n]), (lds_void_a*)piece_dst(slot, i), 16, 0, 0);
      }
    } else {
      for (int i = wave; i < PIECES; i += nw) {
        const bool isv = i >= PT;
        const int j = isv ? i - PT : i;
        const T* base = isv ? V : K;
        const long rs = isv ? p.v_rs : p.k_rs;
        if (TP == 0 || j < 8) {
          const int r = j * 8 + (lane >> 3);
          const int c = (lane & 7) ^ ((r >> 1) & 7);
          __builtin_amdgcn_global_load_lds((glb_void_a*)(base + (long)min(key0 + r, p.Nk - 1) * rs + c * 8), (lds_void_a*)piece_dst(slot, i), 16, 0, 0);
        } else {
          const int r = (j - 8) * (64 / CPR) + lane / CPR;
          const int c = lane % CPR;
          __builtin_amdgcn_global_load_lds((glb_void_a*)(base + (long)min(key0 + r, p.Nk - 1) * rs + 64 + c * 8), (lds_void_a*)piece_dst(slot, i), 16, 0, 0);
        }
      }
    }
  };
  auto wait_own = [&](const bool younger_in_flight) __attribute__((always_inline)) {
    if (!younger_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (my_pieces == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (my_pieces == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (my_pieces == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (my_pieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (my_pieces == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (my_pieces == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  stage(0, 0);
  if (ntiles > 1) stage(1, 1);
  Frag<T> qf[G][NKS];
  short4_t q16[G];
  ag_for<G>([&](auto gc) {
    constexpr int gi = decltype(gc)::value;
    const int q = row_base + gi * nw * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) QLoad<T>::ld(qf[gi][ks], Q + (long)q * p.q_rs + ks * 32 + g * 8, q < p.Nq);
    q16[gi] = (short4_t){0, 0, 0, 0};
    if constexpr (TAIL16) { if (q < p.Nq) q16[gi] = *reinterpret_cast<const short4_t*>(Q + (long)q * p.q_rs + NKS * 32 + g * 4); }
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[gi][ks].v));
    asm volatile("" : "+v"(q16[gi]));
  }

  const int ksw = (l15 >> 1) & 7;
  unsigned koff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) koff[ks] = (unsigned)(l15 * 128 + (((ks * 4 + g) ^ ksw) * 16));
  const unsigned ktail = (unsigned)(MAINB + l15 * TW + g * 16);
  const unsigned ktail16 = (unsigned)(MAINB + l15 * TW + (NKS - 2) * 64 + g * 8);
  const int vkey = g * 4 + (l15 >> 2);
  const int vsw = ((vkey >> 1) & 7) ^ ((l15 & 3) >> 1);
  unsigned voff[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) voff[dt] = (unsigned)(TILE + vkey * 128 + (((dt * 2) ^ vsw) * 16) + (l15 & 1) * 8);
  const unsigned vtail = (unsigned)(TILE + MAINB + vkey * TW + (l15 & 3) * 8);

  float4_t o[G][NDT];
  float m_run[G], l_run[G];
  ag_for<G>([&](auto gc) {
    constexpr int gi = decltype(gc)::value;
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[gi][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    m_run[gi] = -INFINITY; l_run[gi] = 0.f;
  });
  const float cscale = p.scale * 1.4426950408889634f;

  int slot = 0;
  for (int tile = 0; tile < ntiles; ++tile) {
    const int key0 = tile * KT;
    wait_own(tile + 1 < ntiles);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (tile + 2 < ntiles) stage(slot == 0 ? 2 : slot - 1, tile + 2);
    // this tile's LDS addresses: slot base + the per-lane offsets computed once
    const char* S0 = smem + slot * STAGE;
    const char* ka0 = S0 + koff[0];
    const char* ka1 = S0 + koff[1];
    const char* kt_ = S0 + ktail;
    const char* kt16 = S0 + ktail16;
    const char* va[4] = {S0 + voff[0], S0 + voff[1], S0 + voff[2], S0 + voff[3]};
    const char* vt_ = S0 + vtail;
    slot = slot == NST - 1 ? 0 : slot + 1;
    const bool ragged = key0 + KT > p.Nk;

    ag_for<G>([&](auto gc) {
      constexpr int gi = decltype(gc)::value;
      if (row_base + gi * nw * 16 >= p.Nq) return;                      // a group of padding rows only (wave-uniform)
      float4_t sacc[4], tacc[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
        tacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          Frag<T> kf;
          if (ks == 0) kf.v = *reinterpret_cast<const short8_t*>(ka0 + kt * 2048);
          else if (ks == 1) kf.v = *reinterpret_cast<const short8_t*>(ka1 + kt * 2048);
          else kf.v = *reinterpret_cast<const short8_t*>(kt_ + kt * 16 * TW + (ks - 2) * 64);
          mma16(sacc[kt], kf, qf[gi][ks]);
        }
        if constexpr (TAIL16) {      // own accumulator (the SrcC hazard between the 8-pass and 4-pass MFMAs, see attn16_kernel)
          const short4_t k16 = *reinterpret_cast<const short4_t*>(kt16 + kt * 16 * TW);
          mma16_k16<T>(tacc[kt], k16, q16[gi]);
        }
      }
      float sv[16];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[kt * 4 + r] = TAIL16 ? sacc[kt][r] + tacc[kt][r] : sacc[kt][r];
      if (ragged) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + kt * 16 + g * 4 + r >= p.Nk) sv[kt * 4 + r] = -INFINITY;
      }
      float mx = sv[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
      mx = rows4_reduce<true>(mx);
      const float m_new = fmaxf(m_run[gi], mx);
      if (__any(m_new != m_run[gi])) {
        const float alpha = (m_run[gi] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run[gi] - m_new) * cscale);
        l_run[gi] *= alpha;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[gi][dt][r] *= alpha;
        m_run[gi] = m_new;
      }
      const float mc = (m_run[gi] == -INFINITY) ? 0.f : m_run[gi] * cscale;
      float psum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sv[i] = __builtin_amdgcn_exp2f(fmaf(sv[i], cscale, -mc)); psum += sv[i]; }
      l_run[gi] += psum;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float pj[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pj[j] = sv[(kb * 2 + (j >> 2)) * 4 + (j & 3)];
        Frag<T> pf;
        PackP<T>::pack(pf, pj);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          Frag<T> vf;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const char* a = dt < 4 ? va[dt < 4 ? dt : 0] + (kb * 32 + hh * 16) * 128
                                   : vt_ + (kb * 32 + hh * 16) * TW + (dt - 4) * 32;
            const short4_t t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)a);
            vf.v[hh * 4 + 0] = t[0]; vf.v[hh * 4 + 1] = t[1]; vf.v[hh * 4 + 2] = t[2]; vf.v[hh * 4 + 3] = t[3];
          }
          mma16(o[gi][dt], vf, pf);
        }
      }
    });
  }

  ag_for<G>([&](auto gc) {
    constexpr int gi = decltype(gc)::value;
    const int q = row_base + gi * nw * 16 + l15;
    float l = l_run[gi];
    l = rows4_reduce<false>(l);
    const float inv = 1.0f / l;
    if (q < p.Nq) {
      T* O = reinterpret_cast<T*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        T ov[4] = {Elem<T>::from_f(o[gi][dt][0] * inv), Elem<T>::from_f(o[gi][dt][1] * inv), Elem<T>::from_f(o[gi][dt][2] * inv), Elem<T>::from_f(o[gi][dt][3] * inv)};
        *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = *reinterpret_cast<const uint2*>(ov);
      }
    }
  });
}

}  // namespace

#ifdef VLATOUCH_BENCH_BUILD
extern "C" int vt_attn_set_timing(long long* buf) {      // bench build only: device buffer of 4096 x 4 stamps (null = off)
  return hipMemcpyToSymbol(HIP_SYMBOL(d_attn_tbuf), &buf, sizeof(buf)) == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
#endif

static int g_vt_attn16g = -1;     // -1 = read VLATOUCH_ATTN16G at the first launch; 0 = attn16u_kernel everywhere, 1 = grouped-query kernel with G chosen per shape, 3 / 6 = pinned
void vt_attn16g_tune(int value) { g_vt_attn16g = value; }

int vt_attn_launch(const VtAttnParams& p, hipStream_t s) {
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0) return VT_ERR_ARG;
#ifdef VLATOUCH_BENCH_BUILD
  { const char* e = getenv("VLATOUCH_ATTN_ABL"); const int v = e ? atoi(e) : 0; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_attn_abl), &v, sizeof(int), 0, hipMemcpyHostToDevice, s); }
#endif
  if (p.dtype != VT_F32 && p.dtype != VT_BF16 && p.dtype != VT_F16) return VT_ERR_UNSUPPORTED;
  const int epc = p.dtype == VT_F32 ? 4 : 8;
  if (p.q_rs % epc || p.k_rs % epc || p.v_rs % epc || p.q_hs % epc || p.k_hs % epc || p.v_hs % epc) return VT_ERR_ARG;
  int nw = 4, best = 1 << 30;
  for (int w = 4; w <= 8; ++w) {                      // fewest padded query rows; ties -> fewer waves
    const int rows = w * 16, padded = (p.Nq + rows - 1) / rows * rows;
    if (padded < best || (padded == best && p.Nk >= 512 && w == 8)) { best = padded; nw = w; }   // long key sequences: K/V staged once per 128 rows (+3 % on SigLIP)
  }
  const int rows = nw * 16;
  dim3 grid((p.Nq + rows - 1) / rows, p.H, p.B);
  if (p.hd != 0 && p.hd != 64 && p.hd != 96 && p.hd != 80) return VT_ERR_UNSUPPORTED;
  // 16-bit, unmasked, 16-byte-aligned rows: DMA-staged double-buffered tiles (VLATOUCH_ATTN16=0 keeps attn_kernel for A/B)
  static const int a16 = [] { const char* e = getenv("VLATOUCH_ATTN16"); return e ? atoi(e) : 2; }();
  // Grouped-query kernel (attn16g_kernel, G = 6 query groups per wave, the key tiles walked once per block).  Measured (round 5, tools/attn_bench.py,
  // EXPERIMENTS.md): DINOv2-B's 257 tokens as ONE block of 4 waves per (image, head): 48.8 us per layer against 54.4 for attn16u_kernel's three blocks
  // (the six independent groups of a wave overlap their MFMA and softmax streams); the SAME nest on SigLIP's 729 tokens x 80-wide heads is SLOWER
  // (G = 6: 1 459 us against 1 279 — 256 VGPRs + 23 spilled; G = 3, no spills, two blocks per (image, head): 1 308): staging K / V once is not what
  // that launch is short of.  So: 64-wide heads and at most 24 query groups (Nq <= 384) take it, everything else stays on attn16u_kernel.
  // vt_tune(9, v) / VLATOUCH_ATTN16G: 0 = never, 1 = that policy (default), 3 / 6 = every 16-bit unmasked call with G pinned (tests, A/B).
  if (g_vt_attn16g < 0) { const char* e = getenv("VLATOUCH_ATTN16G"); g_vt_attn16g = e ? atoi(e) : 1; }
  const int a16g = g_vt_attn16g;
  // (only under the default variant selection: VLATOUCH_ATTN16=1, the A/B switch for the rolled attn16_kernel, must reach the kernel it names — ADVICE r5)
  if (a16 == 2 && a16g && p.dtype != VT_F32 && !p.kmask && p.o_rs % 4 == 0 && p.Nq >= 128 && (p.hd == 0 || p.hd == 64 || p.hd == 80)) {
    const int need = (p.Nq + 15) / 16;
    int bG = 0, bW = 0;
    if (a16g == 3 || a16g == 6) {                     // pinned: fewest blocks per (image, head), then fewest groups in them, then more waves
      long bcost = 1L << 60;
      for (int w = 4; w <= 8; ++w) {
        const int per = a16g * w, blocks = (need + per - 1) / per;
        const long cost = ((long)blocks << 40) + ((long)(blocks * per - need) << 20) + (8 - w);
        if (cost < bcost) { bcost = cost; bG = a16g; bW = w; }
      }
    } else if (p.hd != 80 && need <= 24) {
      bG = 6; bW = (need + 5) / 6 < 4 ? 4 : (need + 5) / 6;
    }
    if (bG) {
      dim3 gg((need + bG * bW - 1) / (bG * bW), p.H, p.B);
#define VT_A16G(T, HDv) do { if (bG == 6) hipLaunchKernelGGL((attn16g_kernel<T, HDv, 6>), gg, dim3(64 * bW), 0, s, p); \
                             else hipLaunchKernelGGL((attn16g_kernel<T, HDv, 3>), gg, dim3(64 * bW), 0, s, p); } while (0)
      if (p.hd == 80) { if (p.dtype == VT_BF16) VT_A16G(bf16_t, 80); else VT_A16G(half_t, 80); }
      else { if (p.dtype == VT_BF16) VT_A16G(bf16_t, 64); else VT_A16G(half_t, 64); }
#undef VT_A16G
      return vt_check_launch();
    }
  }
  if (a16 && p.dtype != VT_F32 && !p.kmask && p.o_rs % 4 == 0 && p.Nk >= 1) {
#define VT_A16(KERN, T) do { if (p.hd == 96) hipLaunchKernelGGL((KERN<T, 96>), grid, dim3(64 * nw), 0, s, p); \
                       else if (p.hd == 80) hipLaunchKernelGGL((KERN<T, 80>), grid, dim3(64 * nw), 0, s, p); \
                       else hipLaunchKernelGGL((KERN<T, 64>), grid, dim3(64 * nw), 0, s, p); } while (0)
    if (a16 == 2) { if (p.dtype == VT_BF16) VT_A16(attn16u_kernel, bf16_t); else VT_A16(attn16u_kernel, half_t); }
    else { if (p.dtype == VT_BF16) VT_A16(attn16_kernel, bf16_t); else VT_A16(attn16_kernel, half_t); }
#undef VT_A16
    return vt_check_launch();
  }
  if (p.hd == 80) return VT_ERR_UNSUPPORTED;          // 80-wide (padded 72) heads exist only in the 16-bit DMA-staged kernel
  if (p.hd == 96) {
    if (p.dtype == VT_BF16) hipLaunchKernelGGL((attn_kernel<bf16_t, 96>), grid, dim3(64 * nw), 0, s, p);
    else if (p.dtype == VT_F16) hipLaunchKernelGGL((attn_kernel<half_t, 96>), grid, dim3(64 * nw), 0, s, p);
    else hipLaunchKernelGGL((attn_kernel<float, 96>), grid, dim3(64 * nw), 0, s, p);
  } else {
    if (p.dtype == VT_BF16) hipLaunchKernelGGL((attn_kernel<bf16_t, 64>), grid, dim3(64 * nw), 0, s, p);
    else if (p.dtype == VT_F16) hipLaunchKernelGGL((attn_kernel<half_t, 64>), grid, dim3(64 * nw), 0, s, p);
    else hipLaunchKernelGGL((attn_kernel<float, 64>), grid, dim3(64 * nw), 0, s, p);
  }
  return vt_check_launch();
}
