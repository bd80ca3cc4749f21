// vt_gemm_pw.hip — 160 x 128 x 64 tile with the WEIGHT operand streamed global -> VGPR, for 16-bit GEMMs with FROZEN, pre-packed weights
// whose grid is one round (or a few whole rounds) of such tiles: the per-denoise-step Linears of RDT (M = batch x 67 = 2144 rows,
// N = 2048 or 6144, K = 2048; models/rdt/blocks.py:144-183 in the reference).
//
// Why: gemm_ppk_kernel (vt_gemm_ppk.hip) passes BOTH operands through LDS; per k-tile the CU's one LDS port takes 36 KiB of LDS-DMA
// writes and 72 KiB of fragment reads that do not overlap (tools/ubench/fill3.hip), against ~650 clk of MFMA.  The weights never change
// after load, so they are re-packed ONCE (vt_pack_w32) into MFMA fragment order
//     Wp[n/32][k/16][lane = (k%16)/8 * 32 + n%32][k%8]        (a wave's fragment of a 32 x 16 block = one contiguous KiB)
// and each wave streams the fragments of its OWN 32 output columns straight into registers with full-line global_load_dwordx4
// (no LDS write, no LDS read, no duplicate fetch: every byte of W enters the CU once).  Only the activation tile goes through LDS:
//   * 4 waves, one per SIMD (up to 512 VGPRs: accumulators 80, NB weight buffers x 16, two fragment sets x 20); wave w owns the
//     160 x 32 sub-tile at columns 32 w: 5 accumulators of v_mfma_f32_32x32x16 (D[n][m] = W-fragment x A-fragment, so a lane ends with
//     4 consecutive n of one row m);
//   * A k-tiles (160 rows x 128 B = 20 pieces of 1 KiB, XOR-swizzled on the SOURCE address like every LDS-DMA tile of this library)
//     go HBM/L2 -> LDS by DMA into a ring of NB slots; tile t + NB - 1 is issued at the top of iteration t (5 DMA pieces + 4 weight
//     loads per wave, always in this order), the end of iteration t waits — counted, `s_waitcnt vmcnt(9 (NB - 3))` — until tile t + 2
//     has landed, one raw s_barrier per k-tile makes it block-visible;
//   * the weight loads are inline asm (hipcc drains the whole queue with vmcnt(0) at the first use of an ordinary VGPR load issued
//     beside LDS-DMA): the counted wait statement names the buffer it retires as "+v", which is what orders its MFMAs behind it;
//   * fragment reads are software-pipelined one k-step ahead ACROSS the barrier (tile t + 1 became visible one barrier earlier),
//     so a lone wave per SIMD never starts a k-tile with an exposed LDS round trip;
//   * epilogue: the block's 160 x 128 fp32 tile goes through LDS (the ring, XOR-swizzled instead of padded: exactly 80 KiB) and is
//     read back as whole row segments by vt_epi_segment (bias / per-head RMSNorm / activation / column scale / residual), the same
//     arithmetic in the same order as vt_gemm_epilogue.h.
#include <stdlib.h>
#ifdef VLATOUCH_PW_ST
#define VT_EPI_ST_POLICY VLATOUCH_PW_ST
#endif
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_gemm_epilogue.h"
#include "vt_prof.h"
#include "vt_host.h"
#include "vt_kernels.h"
#include "../../include/vlatouch.h"

extern int g_vt_gm;

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) int int4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

constexpr int BM = 160, BN = 128, BK = 64;
constexpr int TMW = BM / 32;                     // 32-row MFMA tiles per wave
constexpr int A_TILE = BM * 128;                 // one k-tile of the activation panel (20 KiB)
constexpr int OPS = BM / 8 / 4 + 4;              // vector-memory operations per wave per k-tile: 5 DMA pieces + 4 weight loads

template <typename T16> __device__ __forceinline__ float16_t mma32(const int4_t w, const int4_t a, const float16_t c);
template <> __device__ __forceinline__ float16_t mma32<bf16_t>(const int4_t w, const int4_t a, const float16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
}
template <> __device__ __forceinline__ float16_t mma32<half_t>(const int4_t w, const int4_t a, const float16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
}

// one KiB of packed weights -> 4 VGPRs per lane; saddr form: uniform 64-bit base + 32-bit lane offset + immediate.  hipcc does not see
// this load: its completion is counted by hand (pw_wait).  FIRST: the base SGPRs may come from a VALU readfirstlane -> 5 wait states.
template <int OFF, bool FIRST>
__device__ __forceinline__ void pw_wload(int4_t& d, const unsigned voff, const char* sbase) {
  if constexpr (FIRST) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
// counted wait that retires weight buffer w[0..3] (and everything older in this wave's queue, i.e. its DMA pieces of that k-tile)
template <int N>
__device__ __forceinline__ void pw_wait(int4_t (&w)[4]) {
#ifdef VLATOUCH_DRAIN_WAITS      // debug build (tools/drain_waits_check.sh)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : [n] "i"(N) : "memory");
#endif
}

template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// ABL (timing-only ablations, tools/gemm_bench_pw.py --abl; results are garbage): 1 = no fragment reads after the prologue, 2 = no weight
// loads after the prologue, 3 = no MFMAs, 4 = no activation DMA after the prologue
// FUSE: 0 = plain, 1 = producer of a fused RMSNorm hand-off (xn_out / xn_part), 2 = its consumer (rs_part) — own instantiations: the plain kernels keep their registers
template <typename T16, typename TC, int NB, int ABL = 0, int FUSE = 0>
__global__ __launch_bounds__(256, 1) void gemm_pw_kernel(const VtGemmParams p, const int tiles_n, const int tiles_per_group, const int total_tiles, const int GM) {
  constexpr int SMEM = NB * A_TILE > BM * BN * 4 ? NB * A_TILE : BM * BN * 4;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];                  // ring of A k-tiles; later the block's fp32 output tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hk = lane >> 5;

  int bid = blockIdx.x;
  if ((total_tiles & 7) == 0) bid = (bid & 7) * (total_tiles >> 3) + (bid >> 3);     // XCD b%8 gets a contiguous band
  const int grp = bid / tiles_per_group;
  const int t_in = bid - grp * tiles_per_group;
  const int tiles_m = tiles_per_group / tiles_n;
  const int sr = t_in / (GM * tiles_n);
  const int gmr = min(GM, tiles_m - sr * GM);
  const int r_in = t_in - sr * GM * tiles_n;
  const int tn = r_in / gmr, tm = sr * GM + (r_in - tn * gmr);
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (long)grp * p.a_gs;
  const int nk = p.K / BK;

  // activation DMA: piece q = 8 tile rows; wave w issues q = w, w+4, ..., w+16.  lane -> (row, chunk position); it fetches the
  // chunk whose swizzled position is its own.  Rows beyond M are clamped (computed, never stored).
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  int asrc[5];
#pragma unroll
  for (int e = 0; e < 5; ++e) {
    const int r = (wave + 4 * e) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    asrc[e] = (int)(((long)(min(m0 + r, p.M - 1) - m0) * p.lda + c * 8) * 2);
  }
  // weight stream of this wave: 32-column block n0/32 + wave, K/16 fragments of 1 KiB back to back
  const char* wbase = reinterpret_cast<const char*>(p.Wp) + ((long)grp * p.w_gs) * 2 + ((long)(n0 / 32 + wave) * (p.K / 16)) * 1024;
  // (the builtin returns int: widen through unsigned, or a low half with bit 31 set sign-extends into the high half)
  wbase = reinterpret_cast<const char*>(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long)wbase >> 32)) << 32) |
                                        (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long)wbase));
  const unsigned wvoff = lane * 16;

  int4_t wb[NB][4];
  auto issue = [&](const int kt, auto bc) {
    constexpr int b = decltype(bc)::value;
    if (ABL != 4 || kt < NB - 1) {
#pragma unroll
      for (int e = 0; e < 5; ++e)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(smem + b * A_TILE + (wave + 4 * e) * 1024), 16, asrc[e], kt * (BK * 2), 0, 0);
    }
    if (ABL != 2 || kt < NB - 1) {
      const char* sb = wbase + (long)kt * 4096;
      pw_wload<0, true>(wb[b][0], wvoff, sb);
      pw_wload<1024, false>(wb[b][1], wvoff, sb);
      pw_wload<2048, false>(wb[b][2], wvoff, sb);
      pw_wload<3072, false>(wb[b][3], wvoff, sb);
    }
  };

  // fragment (32 rows x 16 k) of k-step s of the tile in slot b: lane reads row j*32 + l31, chunk s*2 + hk (swizzled by the row)
  const int xr = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = l31 * 128 + (((s * 2 + hk) ^ xr) * 16);
  auto frags = [&](int4_t (&f)[TMW], const int b, const int s) {
#pragma unroll
    for (int j = 0; j < TMW; ++j) f[j] = *reinterpret_cast<const int4_t*>(smem + b * A_TILE + j * 4096 + foff[s]);
  };

  float16_t acc[TMW];
#pragma unroll
  for (int j = 0; j < TMW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // prologue: tiles 0 .. NB-2 in flight, tiles 0 and 1 landed and block-visible, fragments of (tile 0, step 0) in registers
  static_for<NB - 1>([&](auto bc) { if (decltype(bc)::value < nk) issue(decltype(bc)::value, bc); });
  pw_wait<(NB - 3) * OPS>(wb[0]);
  pw_wait<(NB - 3) * OPS>(wb[1]);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int4_t af[2][TMW];
  frags(af[0], 0, 0);
  if constexpr (ABL == 1) frags(af[1], 0, 1);

  // consumer of a fused RMSNorm: the block's 160 rows x rs_n (sum x^2, sum x) pairs = 160 * rs_n / 2 float4, dealt flat over the 256 threads
  // (float4 q = tid + 256 i: row q / (rs_n / 2), two pairs of that row) — 10 loads per thread at rs_n = 32 instead of 16 on 160 of the threads
  constexpr int RSV = FUSE == 2 ? (BM * 16 + 255) / 256 : 1;
  float4 rsv[RSV];
  auto sub = [&](const int t, auto uc, auto tailc) {
    constexpr int U = decltype(uc)::value;
    constexpr bool TAIL = decltype(tailc)::value;
    if constexpr (!TAIL || U == 0) issue(t + NB - 1, std::integral_constant<int, (U + NB - 1) % NB>{});
    if constexpr (FUSE == 2 && TAIL && U == NB - 2) {
      // requested HERE — behind the last counted wait of the k-loop (the queue is empty, no later wait counts operations) and two k-tiles ahead of
      // the epilogue that needs them, so neither the start of the k-loop nor its end waits for them
      const int h2 = p.rs_n >> 1, nq = BM * h2;
#pragma unroll
      for (int i = 0; i < RSV; ++i) {
        const int q = tid + 256 * i;
        const int row = q / h2, c = q - row * h2;
        rsv[i] = q < nq ? *reinterpret_cast<const float4*>(p.rs_part + ((long)min(m0 + row, p.M - 1) * p.rs_n + 2 * c) * 2) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (ABL != 1) {
        if (s < 3) frags(af[(s + 1) & 1], U, s + 1);
        else if (!TAIL || U < NB - 1) frags(af[0], (U + 1) % NB, 0);
      }
      if constexpr (ABL != 3) {
#pragma unroll
        for (int j = 0; j < TMW; ++j) acc[j] = mma32<T16>(wb[U][s], af[s & 1][j], acc[j]);
      } else {
#pragma unroll
        for (int j = 0; j < TMW; ++j) asm volatile("" : "+v"(af[s & 1][j]), "+v"(wb[U][s]));
      }
    }
    // issue order of the k-tile: MFMA, fragment read, MFMA, ... — every read of step s+1 has five MFMAs (160 clk) to land before its
    // consumer (left alone, hipcc keeps only two reads in flight and a lone wave per SIMD stalls on lgkmcnt before every MFMA)
#pragma unroll
    for (int i = 0; i < 4 * TMW; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
    }
    // tile t + 2: retire it (and this wave's DMA pieces of it), then make every wave's pieces block-visible
    if constexpr (!TAIL) pw_wait<(NB - 3) * OPS>(wb[(U + 2) % NB]);
    else if constexpr (U + 2 < NB) pw_wait<(NB - 3 - U > 0 ? NB - 3 - U : 0) * OPS>(wb[(U + 2) % NB]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  int t = 0;
  for (; t + NB < nk; t += NB) static_for<NB>([&](auto uc) { sub(t + decltype(uc)::value, uc, std::false_type{}); });
  static_for<NB>([&](auto uc) { sub(t + decltype(uc)::value, uc, std::true_type{}); });

  if constexpr (ABL == 5) {      // timing only: no epilogue (the accumulators are kept alive, nothing is stored)
#pragma unroll
    for (int j = 0; j < TMW; ++j) asm volatile("" :: "v"(acc[j]));
    return;
  }
  // ---------------- epilogue: accumulators -> the block's fp32 tile in LDS [160][128], 16-byte columns XOR-swizzled by the row; read back as
  // row segments: 16 lanes cover 64 columns (= one head) of one row, 4 rows per instruction; wave w takes column half w & 1 of rows (w >> 1) * 80 ..
  float* tile = reinterpret_cast<float*>(smem);
  float rs[TMW];
#pragma unroll
  for (int j = 0; j < TMW; ++j) rs[j] = 1.f;
  if constexpr (FUSE == 2) {
    // rstd of the block's 160 rows through the dead ring: every thread folds its float4s to (sum x^2, sum x) of two pairs, one thread per row adds the
    // row's rs_n / 2 entries in a fixed order and hands rstd to the lanes that own the row's accumulators.
    //   mean-square form (timm >= 1.0.9):  rstd = rsqrt(sum x^2 / K + eps)
    //   variance form (timm <= 1.0.8, x not centred):  rstd = rsqrt((sum x^2 - (sum x)^2 / K) / (K - 1) + eps)
    float2* pair = reinterpret_cast<float2*>(tile + BM);
    const int h2 = p.rs_n >> 1;
#pragma unroll
    for (int i = 0; i < RSV; ++i) {
      const int q = tid + 256 * i;
      if (q < BM * h2) {
        // mean-square form: (sum x^2, sum x) of two 64-column groups add; variance form: (M2, sum) of two groups of 64 merge as M2a + M2b + (Sa - Sb)^2 / 128
        const float dS = rsv[i].y - rsv[i].w;
        pair[q] = make_float2(rsv[i].x + rsv[i].z + (p.rs_mode == 2 ? dS * dS * (1.0f / 128.0f) : 0.f), rsv[i].y + rsv[i].w);
      }
    }
    __syncthreads();
    if (tid < BM) {
      float var;
      if (p.rs_mode == 2) {
        // merge the h2 groups of 128 columns in a fixed order: M2 += M2_b + delta^2 n_a n_b / (n_a + n_b), delta = mean_b - mean_a
        float na = 0.f, mean = 0.f, m2 = 0.f;
        for (int c = 0; c < h2; ++c) {
          const float2 v = pair[tid * h2 + c];
          const float delta = v.y * (1.0f / 128.0f) - mean, nt = na + 128.0f;
          mean += delta * (128.0f / nt);
          m2 += v.x + delta * delta * (na * 128.0f / nt);
          na = nt;
        }
        var = m2 / (1.0f / p.rs_inv_k - 1.0f);
      } else {
        float sq = 0.f;
        for (int c = 0; c < h2; ++c) sq += pair[tid * h2 + c].x;
        var = sq * p.rs_inv_k;
      }
      tile[tid] = rsqrtf(var + p.rs_eps);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TMW; ++j) rs[j] = tile[j * 32 + l31];
    __syncthreads();
  }
  const int half = wave & 1, c4 = lane & 15;
  const int ncol0 = n0 + half * 64, n = ncol0 + c4 * 4;
  const bool col_ok = n < p.N;
  const int rbase = (wave >> 1) * (BM / 2) + (lane >> 4);
  TC* Cg = reinterpret_cast<TC*>(p.C) + (long)grp * p.c_gs;
  const TC* Rg = p.residual ? reinterpret_cast<const TC*>(p.residual) + (long)grp * p.r_gs : nullptr;
  // fp32 residual stream (proj / cross proj / fc2 of an RDT block): the lane's 20 residual segments do not depend on the product, so they are
  // requested BEFORE the accumulators go through LDS and land behind that exchange (loaded inside the read-back loop they were 20
  // dependent HBM round trips: the fp32 + residual variant ran 7 us behind the bf16 one)
  constexpr bool PRE = sizeof(TC) == 4;
  float4 rpre[PRE ? BM / 8 : 1];
  const bool use_pre = PRE && Rg != nullptr;
  if constexpr (PRE) {
    if (use_pre) {
#pragma unroll
      for (int it = 0; it < BM / 8; ++it) {
        const int m = min(m0 + rbase + it * 4, p.M - 1);
        rpre[it] = col_ok ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Rg) + (long)m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  // the epilogue's parameter rows (bias, column scale, head-norm gains, the hand-off's norm gains) are requested HERE, with the residual rows and ahead of
  // the prefetch hint below: vmcnt retires in order, so a load issued behind the hint's HBM reads would wait for them
  const float* bias = p.bias ? p.bias + (long)grp * p.bias_gs : nullptr;
  const float* hw = nullptr;
  if (p.hn_w0 && ncol0 < p.hn_c0_end) hw = p.hn_w0;
  else if (p.hn_w1 && ncol0 >= p.hn_c0_end && ncol0 < p.hn_c1_end) hw = p.hn_w1;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b4 = (bias && col_ok) ? *reinterpret_cast<const float4*>(bias + n) : zero4;
  const float4 cs4 = (p.colscale && col_ok) ? *reinterpret_cast<const float4*>(p.colscale + n) : one4;
  const float4 hw4 = hw ? *reinterpret_cast<const float4*>(hw + c4 * 4) : one4;
  float4 g4 = zero4;
  if constexpr (FUSE == 1) { if (col_ok && p.xn_gain) g4 = *reinterpret_cast<const float4*>(p.xn_gain + n); }
  __builtin_amdgcn_sched_barrier(0);
  // ---------------- prefetch hint: this block's share of the NEXT launch's weights, one dword per 64 bytes, fire and forget (retired by the wait the hardware
  // performs before s_endpgm; the values are never used).  Behind every load the epilogue itself consumes: the stream HBM -> Infinity Cache runs under the LDS
  // exchange, the store drain and the kernel boundary.  No ordinary load may follow it in this kernel.
  constexpr int PFN = 8;
  unsigned pfv[PFN];
#pragma unroll
  for (int i = 0; i < PFN; ++i) pfv[i] = 0;
  if (p.pf_ptr) {
    const unsigned nbytes = (unsigned)p.pf_bytes;                                                // (the launcher admits hints below 2 GiB)
    const unsigned share = ((nbytes + (unsigned)total_tiles - 1u) / (unsigned)total_tiles + 63u) & ~63u;   // bytes per block, whole 64-byte units
    const unsigned b0 = blockIdx.x * share;
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.pf_ptr + b0), 0,
                                                                        b0 < nbytes ? (int)min(share, nbytes - b0) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < PFN; ++i)
      if ((unsigned)i * 16384u < share) pfv[i] = __builtin_amdgcn_raw_buffer_load_b32(rsP, (i * 256 + tid) * 64, 0, 0);      // out-of-range lanes: the hardware returns 0
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < TMW; ++j) {
    const int m = j * 32 + l31;
    const float r = FUSE == 2 ? rs[j] : 1.f;                                  // the rows' rstd when this Linear consumes a fused RMSNorm
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int n4 = wave * 8 + rq * 2 + hk;                                  // 16-byte column of n = wave*32 + rq*8 + hk*4 .. +3
      *reinterpret_cast<float4*>(tile + m * BN + ((n4 ^ (m & 7)) * 4)) =
          make_float4(acc[j][rq * 4] * r, acc[j][rq * 4 + 1] * r, acc[j][rq * 4 + 2] * r, acc[j][rq * 4 + 3] * r);
    }
  }
  __syncthreads();
  if (p.act != VT_ACT_NONE) {
#pragma unroll 2
    for (int it = 0; it < BM / 8; ++it) {
      const int row = rbase + it * 4;
      const float4 x = *reinterpret_cast<const float4*>(tile + row * BN + (((half * 16 + c4) ^ (row & 7)) * 4));
      vt_epi_segment<TC, 0, true>(p, x, b4, cs4, hw, hw4, Cg, Rg, m0 + row, n, ncol0, col_ok);
    }
  } else if (FUSE == 1 && PRE && use_pre) {
    // producer of a fused RMSNorm: C = residual + colscale * (product + bias) as below, plus the 16-bit copy C * gain the next Linear reads as its
    // A operand and the (sum of squares, sum) of this row over the wave's 64 columns (all 16 lanes of a row segment take part in the reductions)
    if constexpr (PRE && FUSE == 1) {
      T16* Xn = reinterpret_cast<T16*>(p.xn_out);
      const int pcol = 2 * tn + half, pn = 2 * tiles_n;
#pragma unroll
      for (int it = 0; it < BM / 8; ++it) {
        const int row = rbase + it * 4, m = m0 + row;
        const float4 x = *reinterpret_cast<const float4*>(tile + row * BN + (((half * 16 + c4) ^ (row & 7)) * 4));
        float o[4] = {x.x + b4.x, x.y + b4.y, x.z + b4.z, x.w + b4.w};
        o[0] *= cs4.x; o[1] *= cs4.y; o[2] *= cs4.z; o[3] *= cs4.w;
        o[0] += rpre[it].x; o[1] += rpre[it].y; o[2] += rpre[it].z; o[3] += rpre[it].w;
        const bool ok = m < p.M && col_ok;
        const float q1 = row16_sum(ok ? (o[0] + o[1]) + (o[2] + o[3]) : 0.f);           // the variance form of the consumer also needs the row sum
        float q;
        if (p.rs_mode == 2) {
          // variance form (round 6): the second moment of the 64 columns about THEIR OWN mean — the consumer merges the groups pairwise (Chan et al.), so the
          // row variance never comes from sum x^2 - (sum x)^2 / K, which cancels when |row mean| >> std (the un-fused row-norm kernels centre in two passes too)
          const float mj = q1 * (1.0f / 64.0f);
          const float d0 = o[0] - mj, d1 = o[1] - mj, d2 = o[2] - mj, d3 = o[3] - mj;
          q = row16_sum(ok ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f);
        } else q = row16_sum(ok ? (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]) : 0.f);
        if (ok) {
          vt_epi_st128(reinterpret_cast<float*>(Cg) + (long)m * p.ldc + n, make_float4(o[0], o[1], o[2], o[3]));
          // x * gain is NOT normalised yet (the consumer applies rstd): in IEEE fp16 a residual-stream outlier could leave the range — saturate instead of inf
          constexpr float LIM = std::is_same<T16, half_t>::value ? 65504.0f : 3.0e38f;
          auto sat = [](float v) { return fminf(fmaxf(v, -LIM), LIM); };
          const float xg[4] = {o[0] * g4.x, o[1] * g4.y, o[2] * g4.z, o[3] * g4.w};
          if constexpr (std::is_same<T16, half_t>::value) {      // range guard: the clamp is not silent (vt_rdt_set_range_flag; compute_dtype="auto" re-runs in bf16)
            if (fmaxf(fmaxf(fabsf(xg[0]), fabsf(xg[1])), fmaxf(fabsf(xg[2]), fabsf(xg[3]))) > LIM) vt_range_note(p.range_flag, VT_RANGE_XN_SAT);
          }
          T16 ov[4] = {Elem<T16>::from_f(sat(xg[0])), Elem<T16>::from_f(sat(xg[1])), Elem<T16>::from_f(sat(xg[2])), Elem<T16>::from_f(sat(xg[3]))};
          vt_epi_st64(Xn + (long)m * p.xn_ld + n, *reinterpret_cast<const uint2*>(ov));
        }
        if (c4 == 0 && m < p.M) *reinterpret_cast<float2*>(p.xn_part + ((long)m * pn + pcol) * 2) = make_float2(q, q1);
      }
    }
  } else if (use_pre) {
#pragma unroll
    for (int it = 0; it < BM / 8; ++it) {
      const int row = rbase + it * 4;
      const float4 x = *reinterpret_cast<const float4*>(tile + row * BN + (((half * 16 + c4) ^ (row & 7)) * 4));
      vt_epi_segment<TC, 0, false>(p, x, b4, cs4, hw, hw4, Cg, nullptr, m0 + row, n, ncol0, col_ok, &rpre[PRE ? it : 0]);
    }
  } else {
#pragma unroll 4
    for (int it = 0; it < BM / 8; ++it) {
      const int row = rbase + it * 4;
      const float4 x = *reinterpret_cast<const float4*>(tile + row * BN + (((half * 16 + c4) ^ (row & 7)) * 4));
      vt_epi_segment<TC, 0, false>(p, x, b4, cs4, hw, hw4, Cg, Rg, m0 + row, n, ncol0, col_ok);
    }
  }
#pragma unroll
  for (int i = 0; i < PFN; ++i) asm volatile("" :: "v"(pfv[i]));       // (the prefetch loads are not dead code)
}

// W [N][K] row-major (ldw) -> fragment order: out[((n/32 * K/16 + k/16) * 64 + (k%16)/8 * 32 + n%32) * 8 + k%8]; one thread per 16-byte chunk
__global__ void pack_w32_kernel(const uint16_t* __restrict__ W, const long ldw, uint16_t* __restrict__ out, const int N, const int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // chunk index in the OUTPUT order
  const long nchunks = (long)N * K / 8;
  if (i >= nchunks) return;
  const int lane = (int)(i & 63);
  const long frag = i >> 6;
  const int ks = (int)(frag % (K / 16));
  const int nt = (int)(frag / (K / 16));
  const int n = nt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8;
  *reinterpret_cast<uint4*>(out + i * 8) = *reinterpret_cast<const uint4*>(W + (long)n * ldw + k);
}

}  // namespace

#ifdef VLATOUCH_BENCH_BUILD
static int g_vt_pw_abl = 0;     // vt_tune(5, k): timing-only ablation k of gemm_pw_kernel<bf16, bf16, 4> (0 = off)
#endif
static int g_vt_pw_nb = 0;      // ring depth 4 | 8 (VLATOUCH_PW_NB, vt_tune(1, .)); 0 = default
static int g_vt_pw_on = 1;      // VLATOUCH_PW=0 / vt_tune(2, 0) disables the kernel (A/B against gemm_ppk_kernel / gemm_pp256d_kernel)

bool vt_gemm_pw_eligible(const VtGemmParams& p) {
  static const bool init = [] { const char* e = getenv("VLATOUCH_PW"); const char* nb = getenv("VLATOUCH_PW_NB"); if (nb) g_vt_pw_nb = atoi(nb); if (e) g_vt_pw_on = atoi(e) != 0; return true; }();
  (void)init;
  if (!g_vt_pw_on || !p.Wp || !vt_gemm_fast_eligible(p) || p.cmap || p.groups != 1) return false;
  if (p.N % BN || p.K % (8 * BK) || p.lda >= (1 << 21)) return false;
  const long tiles = (long)((p.M + BM - 1) / BM) * (p.N / BN) * p.groups;
  // one round of the 256 CUs, or several rounds each at least 7/8 full
  const long rounds = (tiles + 255) / 256;
  return tiles >= 100 && tiles * 8 >= rounds * 256 * 7;
}

int vt_gemm_pw_launch(const VtGemmParams& p, hipStream_t s) {
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  const int per_group = tiles_n * tiles_m, total = per_group * p.groups;
  const int gm = g_vt_gm > 0 ? g_vt_gm : 4;
  VtProfScope prof(3, p, s);
  const bool c16 = p.c_dtype != VT_F32;
#define VT_PW_GO(T16, TC) do { if (g_vt_pw_nb == 8) hipLaunchKernelGGL((gemm_pw_kernel<T16, TC, 8>), dim3(total), dim3(256), 0, s, p, tiles_n, per_group, total, gm); \
                               else hipLaunchKernelGGL((gemm_pw_kernel<T16, TC, 4>), dim3(total), dim3(256), 0, s, p, tiles_n, per_group, total, gm); } while (0)
#ifdef VLATOUCH_BENCH_BUILD      // timing-only ablations (garbage results): compiled only into a bench build (make DEFS=-DVLATOUCH_BENCH_BUILD)
  if (g_vt_pw_abl && p.a_dtype == VT_BF16 && c16) {
    const dim3 g(total), b(256);
    if (g_vt_pw_abl == 1) hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, bf16_t, 4, 1>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    else if (g_vt_pw_abl == 2) hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, bf16_t, 4, 2>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    else if (g_vt_pw_abl == 3) hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, bf16_t, 4, 3>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    else if (g_vt_pw_abl == 4) hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, bf16_t, 4, 4>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    else hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, bf16_t, 4, 5>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    return vt_check_launch();
  }
#endif
  if (p.xn_out || p.rs_part) {        // fused RMSNorm hand-off (vt_gemm.h): ring depth 4 only
    const dim3 g(total), b(256);
    if (p.xn_out && (c16 || !p.residual || p.act != VT_ACT_NONE || p.hn_w0 || p.hn_w1)) return VT_ERR_UNSUPPORTED;
    if (p.xn_out && p.rs_part) return VT_ERR_UNSUPPORTED;
    if (p.xn_out) {
      if (p.a_dtype == VT_BF16) hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, float, 4, 0, 1>), g, b, 0, s, p, tiles_n, per_group, total, gm);
      else hipLaunchKernelGGL((gemm_pw_kernel<half_t, float, 4, 0, 1>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    } else if (!c16) {
      return VT_ERR_UNSUPPORTED;        // consumers are the 16-bit-output Linears (qkv, cross q, fc1); an fp32-output consumer would need 260 VGPRs (no second block per CU)
    } else if (p.a_dtype == VT_BF16) {
      hipLaunchKernelGGL((gemm_pw_kernel<bf16_t, bf16_t, 4, 0, 2>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    } else {
      hipLaunchKernelGGL((gemm_pw_kernel<half_t, half_t, 4, 0, 2>), g, b, 0, s, p, tiles_n, per_group, total, gm);
    }
    return vt_check_launch();
  }
  if (p.a_dtype == VT_BF16) { if (c16) VT_PW_GO(bf16_t, bf16_t); else VT_PW_GO(bf16_t, float); }
  else { if (c16) VT_PW_GO(half_t, half_t); else VT_PW_GO(half_t, float); }
#undef VT_PW_GO
  return vt_check_launch();
}

void vt_unet_fused_tune(int on);

extern "C" int vt_tune(int knob, int value) {
  VtGemmParams dummy{};
  (void)vt_gemm_pw_eligible(dummy);          // environment defaults are read before the first explicit setting
  if (knob == 1 && (value == 0 || value == 4 || value == 8)) { g_vt_pw_nb = value; return VT_OK; }
  if (knob == 2) { g_vt_pw_on = value != 0; return VT_OK; }
#ifdef VLATOUCH_BENCH_BUILD
  if (knob == 5 && value >= 0 && value <= 5) { g_vt_pw_abl = value; return VT_OK; }
#else
  if (knob == 5) return value == 0 ? VT_OK : vt_fail(VT_ERR_UNSUPPORTED, "vt_tune(5, .): the timing-only ablations exist only in a bench build (make DEFS=-DVLATOUCH_BENCH_BUILD)");
#endif
  if (knob == 6) { vt_attn_kvt_tune(value); return VT_OK; }
  if (knob == 7) { vt_unet_fused_tune(value); return VT_OK; }
  if (knob == 8) { vt_gemm_pt_tune(value); return VT_OK; }
  if (knob == 9 && (value == 0 || value == 1 || value == 3 || value == 6)) { vt_attn16g_tune(value); return VT_OK; }
  if (knob == 3 || knob == 4) { vt_gemm_pws_tune(knob, value); return VT_OK; }
  return vt_fail(VT_ERR_ARG, "vt_tune: unknown knob %d / value %d", knob, value);
}

extern "C" int vt_pack_w32(const void* W, long ldw, void* out, int N, int K, vt_stream_t stream) {
  if (!W || !out || N <= 0 || K <= 0 || N % 32 || K % 16 || ldw % 8) return vt_fail(VT_ERR_ARG, "vt_pack_w32: N %% 32, K %% 16, ldw %% 8 must be 0");
  const long nchunks = (long)N * K / 8;
  hipLaunchKernelGGL(pack_w32_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)W, ldw, (uint16_t*)out, N, K);
  return vt_check_launch();
}
