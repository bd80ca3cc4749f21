// vt_gemm_ppk.hip — 160 x 128 x 64 tile with IN-BLOCK SPLIT-K ping-pong, for 16-bit GEMMs whose whole grid is ONE round of
// such tiles (120..256 blocks): the per-denoise-step Linears of RDT (M = batch x 67 = 2144 rows, N = 2048 -> 14 x 16 = 224 tiles).
// These GEMMs are bound by the LDS-DMA issue rate of a CU (~27 B/clk, vt_gemm_pp.hip), i.e. by the operand bytes a CU has to
// pull in; 64 x 128 tiles (the only 128-column tile that fills the chip at this M) pull 24 KiB per 1.05 MFLOP.  One 160 x 128
// tile per CU pulls 36 KiB per 2.6 MFLOP — 1.6x fewer bytes per flop — but a lone block per CU has nobody to overlap with, so:
//   * 8 waves = 2 groups x (2 x 2) waves.  BOTH groups cover the whole 160 x 128 tile (wave tile 80 x 64) but group g takes the
//     k-tiles of parity g, into its own accumulators and its own pair of LDS buffers (2 x 2 x 36 KiB = 144 KiB);
//   * the groups run the same code staggered by one barrier: while one group issues the 18 fragment reads of its k-tile and the
//     9 DMA pieces per wave of its next one, the other runs its 40 MFMAs per wave (s_setprio 1);
//   * the DMA of a group's next k-tile has a whole MFMA phase to land: the wave waits for its own pieces at the end of that
//     phase, the closing barrier makes them block-visible, the following memory phase reads them (RAW); a buffer is restaged
//     two of the group's k-tiles after its last read (WAR);
//   * at the end the groups swap halves of their accumulators through LDS (80 KiB, lane-contiguous): group 0 finishes row tiles
//     0..2 of every wave tile, group 1 row tiles 3..4 (a + b is commutative: the result does not depend on timing), and both
//     run the shared epilogue on their part.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_gemm_epilogue.h"
#include "vt_prof.h"

extern int g_vt_gm;

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BM = 160, BN = 128, BK = 64;
constexpr int TM = 5;                            // 16-row MFMA tiles per wave along M (wave tile 80 x 64)
constexpr int BUF_BYTES = (BM + BN) * 128;       // one k-tile: A rows 0..159 then B rows 0..127, 128 B each (36 KiB)

template <typename T16, typename TC>
__global__ __launch_bounds__(512, 2) void gemm_ppk_kernel(const VtGemmParams p, const int tiles_n, const int tiles_per_group, const int total_tiles, const int GM) {
  constexpr int XCH_BYTES = 4 * (4 * TM * 4) * 64 * 4;                       // accumulator hand-over area (80 KiB)
  constexpr int SMEM = 4 * BUF_BYTES > XCH_BYTES + 8 * EP_BYTES ? 4 * BUF_BYTES : XCH_BYTES + 8 * EP_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];                 // [group][stage] operand buffers; later hand-over area + 8 patches
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp_k = wave >> 2, wq = wave & 3;      // k-parity group, wave inside the group
  const int wm = wq >> 1, wn = wq & 1;
  const int g = lane >> 4, l15 = lane & 15;

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
  const uint16_t* W = reinterpret_cast<const uint16_t*>(p.W) + (long)grp * p.w_gs;

  // DMA pieces of a k-tile: piece q = 8 tile rows (A rows for q < 20, then B rows); wave wq of the group issues q = wq, wq+4, ...
  // lane -> (row, chunk position); it fetches the chunk whose swizzled position is its own.  Rows beyond M / N are clamped.
  const uint16_t* src[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const int q = wq + 4 * e;
    const int r = (q < BM / 8 ? q * 8 : (q - BM / 8) * 8) + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    src[e] = q < BM / 8 ? A + (long)min(m0 + r, p.M - 1) * p.lda + c * 8 : W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 8;
  }
  char* gbase = smem + grp_k * 2 * BUF_BYTES;
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int e = 0; e < 9; ++e)
      __builtin_amdgcn_global_load_lds((glb_void*)(src[e] + (long)kt * BK), (lds_void*)(gbase + buf * BUF_BYTES + (wq + 4 * e) * 1024), 16, 0, 0);
  };

  float4_t acc[4][TM];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int ns = (nk - grp_k + 1) / 2;            // k-tiles of this group: kt = 2*s + grp_k
  const int ns_max = (nk + 1) / 2;                // both groups run the same number of barrier pairs
  if (ns > 0) stage(0, grp_k);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp_k == 1) __builtin_amdgcn_s_barrier();   // stagger group 1 by one barrier (group 0 pays it back after the loop)

  const int arow = wm * (BM / 2) + l15, brow = wn * 64 + l15;
  for (int s = 0; s < ns_max; ++s) {
    const bool live = s < ns;                     // group-uniform: the group with fewer k-tiles idles through its last pair
    const char* As = gbase + (s & 1) * BUF_BYTES;
    const char* Bs = As + BM * 128;
    Frag<T16> af[TM][2], wf[4][2];
    // ---- memory phase: this k-tile's fragments, next k-tile's DMA
    if (live) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lds_frag(wf[i][ks], Bs, brow + i * 16, ks * 4 + g);
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lds_frag(af[j][ks], As, arow + j * 16, ks * 4 + g);
      if (s + 1 < ns) stage((s + 1) & 1, 2 * (s + 1) + grp_k);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMA phase
    if (live) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[i][ks], af[j][ks]);
      __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next k-tile (issued a whole phase ago)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp_k == 0) __builtin_amdgcn_s_barrier();   // pay back the stagger
  __syncthreads();                                // every fragment read is done: the buffers become the hand-over area

  // ---- the groups swap halves through LDS ([wq][register][lane] floats, lane-contiguous: conflict-free; 4 x 80 x 64 x 4 B =
  // 80 KiB): group 0 ends up with the summed row tiles 0..2 of every wave tile, group 1 with row tiles 3..4, and BOTH run the
  // shared epilogue on their part (it is a third of this kernel's time at K = 2048: bias / norm / activation / residual on
  // 160 x 128 outputs with one block per CU and nothing else to overlap it with).  a + b is commutative: the result does not
  // depend on which group adds.
  constexpr int T0 = 3, T1 = TM - T0;              // row tiles finished by group 0 / group 1
  float* xch = reinterpret_cast<float*>(smem) + (long)wq * (4 * TM * 4) * 64 + lane;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const bool mine = grp_k == 0 ? (j < T0) : (j >= T0);
      if (!mine) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[((i * TM + j) * 4 + r) * 64] = acc[i][j][r];
      }
    }
  __syncthreads();
  float* ep = reinterpret_cast<float*>(smem + XCH_BYTES + wave * EP_BYTES);      // private patch beyond the hand-over area (others may still read it)
  if (grp_k == 0) {
    float4_t part[4][T0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < T0; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[i][j][r] = acc[i][j][r] + xch[((i * TM + j) * 4 + r) * 64];
    vt_gemm_epilogue<TC, T0, 0>(p, part, ep, grp, m0 + wm * (BM / 2), n0 + wn * 64, lane);
  } else {
    float4_t part[4][T1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < T1; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[i][j][r] = acc[i][T0 + j][r] + xch[((i * TM + T0 + j) * 4 + r) * 64];
    vt_gemm_epilogue<TC, T1, 0>(p, part, ep, grp, m0 + wm * (BM / 2) + T0 * 16, n0 + wn * 64, lane);
  }
}

}  // namespace

bool vt_gemm_ppk_eligible(const VtGemmParams& p) {
  if (!vt_gemm_fast_eligible(p) || p.cmap) return false;
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.groups;
  return tiles >= 100 && tiles <= 256 && p.K >= 512;     // one round of 160 x 128 tiles over the 256 CUs
}

int vt_gemm_ppk_launch(const VtGemmParams& p, hipStream_t s) {
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int per_group = tiles_n * tiles_m, total = per_group * p.groups;
  const int gm = g_vt_gm > 0 ? g_vt_gm : 4;
  VtProfScope prof(3, p, s);
#define VT_PPK_GO(T16, TC) hipLaunchKernelGGL((gemm_ppk_kernel<T16, TC>), dim3(total), dim3(512), 0, s, p, tiles_n, per_group, total, gm)
  const bool c16 = p.c_dtype != VT_F32;
  if (p.a_dtype == VT_BF16) { if (c16) VT_PPK_GO(bf16_t, bf16_t); else VT_PPK_GO(bf16_t, float); }
  else { if (c16) VT_PPK_GO(half_t, half_t); else VT_PPK_GO(half_t, float); }
#undef VT_PPK_GO
  return vt_check_launch();
}
