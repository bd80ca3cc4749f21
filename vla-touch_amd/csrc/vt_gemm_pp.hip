// vt_gemm_pp.hip — 256 x 256 x 64 "ping-pong" tile of the large-GEMM path (16-bit x 16-bit -> fp32 on
// v_mfma_f32_16x16x32_{bf16,f16}), for GEMMs with several rounds of 256-square tiles (the cached-condition K/V projections and
// the image adaptor of RDT).  The 128-square kernel of vt_gemm_fast.hip tops out at the L2 -> LDS fill rate (measured
// ~27 B/clk/CU, i.e. ~930 TF/s at its 64 flop/B); a 256-square tile needs half the fill bytes per flop, but with one block
// per CU nothing hides its memory phases unless the block does it itself:
//   * 8 waves = 2 groups x 4 (one wave of each group per SIMD).  Group g owns rows g*128 .. +127, wave (g, wn) the 128 x 64
//     sub-tile at columns wn*64.  The groups run the SAME code staggered by one barrier, so on every SIMD one wave is in its
//     MFMA cluster (s_setprio 1) while the other issues its LDS reads and DMA: the matrix pipe and the memory pipes alternate
//     owners instead of idling in turn.
//   * a k-tile is 4 phases of 16 MFMAs (one 64 x 32 quadrant of the wave's sub-tile x K=64): [LDS fragment reads + 2 DMA
//     pieces] barrier [16 MFMAs] barrier.  Quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0) reads 12 / 4 / 8 / 0 fragments.
//   * the operand tile of k-tile t+1 is staged in 4 units of 16 KiB, one per phase, in the order the phases of t+1 consume them
//     (U0 = A rows of every wave's first half, U1 / U2 = B rows of the first / second column half, U3 = A second half).  Waits
//     are COUNTED: `s_waitcnt vmcnt(4)` after each issue leaves the two youngest units in flight, and the barriers are raw
//     s_barrier (a __syncthreads() would drain the DMA queue).  A unit is read one phase after the wait + barrier that retires
//     it (RAW: the wait is per wave, the barrier makes it block-wide, also across the stagger), and its LDS slot is restaged
//     no sooner than 4 phases after its last read (WAR).
// LDS: 2 buffers x (A 256 x 128 B + B 256 x 128 B) = 128 KiB, rows XOR-swizzled exactly as in vt_gemm_fast.hip (lane-linear
// DMA image, swizzle on the source address).  Tile order, XCD banding and the epilogue are shared with that kernel.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_gemm_epilogue.h"
#include "vt_prof.h"

extern int g_vt_gm;

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int BUF_BYTES = (BM + BN) * 128;      // one k-tile: A rows 0..255 then B rows 0..255, 128 B each

#ifdef VLATOUCH_DRAIN_WAITS      // debug build (tools/drain_waits_check.sh): counted waits drain the queue
#define VT_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define VT_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif

template <typename T16, typename TC, int CMAP>
__global__ __launch_bounds__(512, 2) void gemm_pp256_kernel(const VtGemmParams p, const int tiles_n, const int tiles_per_group, const int total_tiles, const int GM) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // group (row half), column quarter
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

  // DMA pieces: a unit is 128 tile rows = 16 pieces of 8 rows; this wave issues pieces 2*wave and 2*wave+1 of every unit.
  //   A units (U0: sel 0, U3: sel 1): rows with (r & 64) == sel*64 -> r = (q>>3)*128 + sel*64 + (q&7)*8 + lane/8
  //   B units (U1: sel 0, U2: sel 1): rows with (r & 32) == sel*32 -> r = (q>>2)*64  + sel*32 + (q&3)*8 + lane/8
  // lane -> (row, chunk position); it fetches the chunk whose swizzled position is its own.  Rows beyond M / N are clamped.
  // The pieces are buffer loads (SGPR resource of the tile's A / W row block + one 32-bit VGPR offset per lane + the k offset in
  // an SGPR): the per-piece issue cost of the LDS-DMA is what bounds this kernel, and a 32-bit offset is half the address traffic
  // of a 64-bit flat pointer per lane.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int src[4][2];
  int dst[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = 2 * wave + e;
      const bool isA = (u == 0 || u == 3);
      const int sel = (u == 2 || u == 3) ? 1 : 0;
      const int r8 = isA ? ((q >> 3) * 128 + sel * 64 + (q & 7) * 8) : ((q >> 2) * 64 + sel * 32 + (q & 3) * 8);
      const int r = r8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      src[u][e] = isA ? (int)(((long)(min(m0 + r, p.M - 1) - m0) * p.lda + c * 8) * 2) : (int)(((long)(min(n0 + r, p.N - 1) - n0) * p.ldw + c * 8) * 2);
      dst[u][e] = (isA ? 0 : BM * 128) + r8 * 128;
    }
  auto stage = [&](int u, int buf, int kt) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds((u == 0 || u == 3) ? rsA : rsW, (lds_void*)(smem + buf * BUF_BYTES + dst[u][e]), 16, src[u][e], kt * (BK * 2), 0, 0);
  };

  float4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0, 0); stage(1, 0, 0); stage(2, 0, 0); stage(3, 0, 0);
  VT_WAIT_VM(4);                                   // U0, U1 of k-tile 0
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // stagger group 1 by one barrier (group 0 pays it back after the loop)

  const int arow = wm * 128 + l15, brow = wn * 64 + l15;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    const char* As = smem + cur * BUF_BYTES;
    const char* Bs = As + BM * 128;
    Frag<T16> a0[4][2], a1[4][2], b0[2][2], b1[2][2];

#define VT_PP_MEM_END(u)                                            \
    if (more) { stage(u, cur ^ 1, kt + 1); VT_WAIT_VM(4); }         \
    else { VT_WAIT_VM(0); }                                         \
    __builtin_amdgcn_sched_barrier(0);                              \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_sched_barrier(0);
#define VT_PP_MMA(AF, BF, ah, bh)                                   \
    __builtin_amdgcn_s_setprio(1);                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) mma16(acc[(bh) * 2 + i][(ah) * 4 + j], BF[i][ks], AF[j][ks]); \
    __builtin_amdgcn_s_setprio(0);                                  \
    __builtin_amdgcn_sched_barrier(0);                              \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_sched_barrier(0);

    // phase 0: quadrant (A0, B0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(b0[i][ks], Bs, brow + i * 16, ks * 4 + g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(a0[j][ks], As, arow + j * 16, ks * 4 + g);
    VT_PP_MEM_END(0)
    VT_PP_MMA(a0, b0, 0, 0)
    // phase 1: quadrant (A0, B1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(b1[i][ks], Bs, brow + 32 + i * 16, ks * 4 + g);
    VT_PP_MEM_END(1)
    VT_PP_MMA(a0, b1, 0, 1)
    // phase 2: quadrant (A1, B1)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(a1[j][ks], As, arow + 64 + j * 16, ks * 4 + g);
    VT_PP_MEM_END(2)
    VT_PP_MMA(a1, b1, 1, 1)
    // phase 3: quadrant (A1, B0), no reads
    VT_PP_MEM_END(3)
    VT_PP_MMA(a1, b0, 1, 0)
#undef VT_PP_MEM_END
#undef VT_PP_MMA
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();       // pay back the stagger
  __syncthreads();                                 // every fragment read is done: the buffers become the epilogue patches

  vt_gemm_epilogue<TC, 8, CMAP>(p, acc, reinterpret_cast<float*>(smem + wave * EP_BYTES), grp, m0 + wm * 128, n0 + wn * 64, lane);
}

template <typename T16, typename TC, int CMAP>
__global__ __launch_bounds__(512, 2) void gemm_pp256d_kernel(const VtGemmParams p, const int tiles_n, const int tiles_per_group, const int total_tiles, const int GM) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // group (row half), column quarter
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

  // DMA pieces: a unit is 128 tile rows = 16 pieces of 8 rows; this wave issues pieces 2*wave and 2*wave+1 of every unit.
  //   A units (U0: sel 0, U3: sel 1): rows with (r & 64) == sel*64 -> r = (q>>3)*128 + sel*64 + (q&7)*8 + lane/8
  //   B units (U1: sel 0, U2: sel 1): rows with (r & 32) == sel*32 -> r = (q>>2)*64  + sel*32 + (q&3)*8 + lane/8
  // lane -> (row, chunk position); it fetches the chunk whose swizzled position is its own.  Rows beyond M / N are clamped.
  // The pieces are buffer loads (SGPR resource of the tile's A / W row block + one 32-bit VGPR offset per lane + the k offset in
  // an SGPR): the per-piece issue cost of the LDS-DMA is what bounds this kernel, and a 32-bit offset is half the address traffic
  // of a 64-bit flat pointer per lane.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
  int src[4][2];
  int dst[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = 2 * wave + e;
      const bool isA = (u == 0 || u == 3);
      const int sel = (u == 2 || u == 3) ? 1 : 0;
      const int r8 = isA ? ((q >> 3) * 128 + sel * 64 + (q & 7) * 8) : ((q >> 2) * 64 + sel * 32 + (q & 3) * 8);
      const int r = r8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      src[u][e] = isA ? (int)(((long)(min(m0 + r, p.M - 1) - m0) * p.lda + c * 8) * 2) : (int)(((long)(min(n0 + r, p.N - 1) - n0) * p.ldw + c * 8) * 2);
      dst[u][e] = (isA ? 0 : BM * 128) + r8 * 128;
    }
  auto stage = [&](int u, int buf, int kt) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds((u == 0 || u == 3) ? rsA : rsW, (lds_void*)(smem + buf * BUF_BYTES + dst[u][e]), 16, src[u][e], kt * (BK * 2), 0, 0);
  };

  float4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // DEEP schedule: a unit's LDS slot is re-staged two phases after its last read (with the unit of the tile TWO ahead that lives
  // in the same slot) instead of idling until the next k-tile: issue order = consumption order U0 U1 U2 U3 of tile 0, 1, 2, ...;
  // unit n = (tile n/4, U n%4) sits in buffer (n/4)&1; phase P = 4*kt + ph reads units <= P+1 (ph 0: U0 U1, ph 1: U2, ph 2: U3), issues
  // unit P+6 and then waits until unit P+2 has landed (the next phase's reads) with the FOUR younger units (64 KiB) still in
  // flight: `s_waitcnt vmcnt(8)`, twice the lead of the one-tile-ahead schedule above at the same 128 KiB of LDS.
  //   WAR: unit P+6 overwrites the slot of unit P-2, last read in phase P-2 or earlier by both (staggered) groups: two barriers ago.
  //   RAW: unit P+2 is retired by every issuing wave's counted wait before the barrier that ends the memory half of phase P; its
  //        first read is in phase P+1.
  const int nk = p.K / BK;
  const int NU = 4 * nk;
#pragma unroll
  for (int n = 0; n < 6; ++n)
    if (n < NU) stage(n & 3, (n >> 2) & 1, n >> 2);
  if (NU > 4) { VT_WAIT_VM(8); } else { VT_WAIT_VM(4); }      // units 0, 1 landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // stagger group 1 by one barrier (group 0 pays it back after the loop)

  const int arow = wm * 128 + l15, brow = wn * 64 + l15;
  auto tile = [&](const int kt, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    const int cur = kt & 1;
    const char* As = smem + cur * BUF_BYTES;
    const char* Bs = As + BM * 128;
    Frag<T16> a0[4][2], a1[4][2], b0[2][2], b1[2][2];
    auto mem_end = [&](auto ph_tag) {
      constexpr int ph = decltype(ph_tag)::value;
      constexpr int U = (ph + 2) & 3, DT = (ph + 6) >> 2;      // unit P+6 = (tile kt + DT, U): compile-time unit -> register-resident offsets
      const int P = 4 * kt + ph, n = P + 6;
      if (STEADY) {
        stage(U, (kt + DT) & 1, kt + DT);
        VT_WAIT_VM(8);
      } else {
        if (n < NU) stage(U, (kt + DT) & 1, kt + DT);
        const int fly = min(4, NU - 1 - (P + 2));              // units younger than P+2 that have been issued
        if (fly >= 4) VT_WAIT_VM(8);
        else if (fly == 3) VT_WAIT_VM(6);
        else if (fly == 2) VT_WAIT_VM(4);
        else if (fly == 1) VT_WAIT_VM(2);
        else VT_WAIT_VM(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
#define VT_PPD_MMA(AF, BF, ah, bh)                                  \
    __builtin_amdgcn_s_setprio(1);                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) mma16(acc[(bh) * 2 + i][(ah) * 4 + j], BF[i][ks], AF[j][ks]); \
    __builtin_amdgcn_s_setprio(0);                                  \
    __builtin_amdgcn_sched_barrier(0);                              \
    __builtin_amdgcn_s_barrier();                                   \
    __builtin_amdgcn_sched_barrier(0);
    // phase 0: quadrant (A0, B0)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(b0[i][ks], Bs, brow + i * 16, ks * 4 + g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(a0[j][ks], As, arow + j * 16, ks * 4 + g);
    mem_end(std::integral_constant<int, 0>{});
    VT_PPD_MMA(a0, b0, 0, 0)
    // phase 1: quadrant (A0, B1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(b1[i][ks], Bs, brow + 32 + i * 16, ks * 4 + g);
    mem_end(std::integral_constant<int, 1>{});
    VT_PPD_MMA(a0, b1, 0, 1)
    // phase 2: quadrant (A1, B1)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(a1[j][ks], As, arow + 64 + j * 16, ks * 4 + g);
    mem_end(std::integral_constant<int, 2>{});
    VT_PPD_MMA(a1, b1, 1, 1)
    // phase 3: quadrant (A1, B0), no reads
    mem_end(std::integral_constant<int, 3>{});
    VT_PPD_MMA(a1, b0, 1, 0)
#undef VT_PPD_MMA
  };
  int kt = 0;
  for (; kt + 2 < nk; ++kt) tile(kt, std::true_type{});      // every phase of these tiles issues a unit (4*kt + 3 + 6 < 4*nk)
  for (; kt < nk; ++kt) tile(kt, std::false_type{});
  if (wm == 0) __builtin_amdgcn_s_barrier();       // pay back the stagger
  __syncthreads();                                 // every fragment read is done: the buffers become the epilogue patches

  vt_gemm_epilogue<TC, 8, CMAP>(p, acc, reinterpret_cast<float*>(smem + wave * EP_BYTES), grp, m0 + wm * 128, n0 + wn * 64, lane);
}

}  // namespace

bool vt_gemm_pp_eligible(const VtGemmParams& p) { return vt_gemm_pp_shape(p) || vt_gemm_pt_extra_shape(p); }

// the shapes gemm_pp256d_kernel is good at (the persistent kernel of vt_gemm_pt.hip takes these and a few more, vt_gemm_pt_extra_shape)
bool vt_gemm_pp_shape(const VtGemmParams& p) {
  if (!vt_gemm_fast_eligible(p)) return false;
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.groups;
  if (p.lda >= (1 << 21) || p.ldw >= (1 << 21) || p.K >= (1 << 24)) return false;   // 32-bit buffer offsets inside a 256-row block
  if (tiles256 >= 192 && tiles256 <= 256 && p.K >= 1024) return true;   // one well-filled round (RDT qkv: 9 x 24 tiles): 22 % faster than 128-col tiles
  return tiles256 >= 512 && p.K >= 512;            // at least two rounds of 256-square tiles over the 256 CUs
}

int vt_gemm_pp_launch(const VtGemmParams& p, hipStream_t s) {
  if (vt_gemm_pt_eligible(p)) return vt_gemm_pt_launch(p, s);      // persistent tile walk + in-register epilogue folded into the main loop (vt_gemm_pt.hip)
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int per_group = tiles_n * tiles_m, total = per_group * p.groups;
  const int gm = g_vt_gm > 0 ? g_vt_gm : 8;
  VtProfScope prof(2, p, s);
  static const int deep = [] { const char* e = getenv("VLATOUCH_PP_DEEP"); return e ? atoi(e) : 1; }();   // 0 = one-tile-ahead schedule (A/B)
#define VT_PP_GO(T16, TC, CM) do { if (deep) hipLaunchKernelGGL((gemm_pp256d_kernel<T16, TC, CM>), dim3(total), dim3(512), 0, s, p, tiles_n, per_group, total, gm); \
                                   else hipLaunchKernelGGL((gemm_pp256_kernel<T16, TC, CM>), dim3(total), dim3(512), 0, s, p, tiles_n, per_group, total, gm); } while (0)
  const bool c16 = p.c_dtype != VT_F32;
  if (p.cmap == 3 && p.a_dtype == VT_F16) VT_PP_GO(half_t, half_t, 3);
  else if (p.cmap == 1) VT_PP_GO(bf16_t, bf16_t, 1);
  else if (p.cmap == 2) VT_PP_GO(bf16_t, bf16_t, 2);
  else if (p.cmap == 3) VT_PP_GO(bf16_t, bf16_t, 3);
  else if (p.a_dtype == VT_BF16) { if (c16) VT_PP_GO(bf16_t, bf16_t, 0); else VT_PP_GO(bf16_t, float, 0); }
  else { if (c16) VT_PP_GO(half_t, half_t, 0); else VT_PP_GO(half_t, float, 0); }
#undef VT_PP_GO
  return vt_check_launch();
}
