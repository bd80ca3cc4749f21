// vt_gemm_pt.hip — PERSISTENT 256 x 256 x 64 ping-pong tile (round 4).  Same wave layout, LDS image, XOR swizzle and deep-prefetch
// schedule as gemm_pp256d_kernel (vt_gemm_pp.hip: 8 waves = 2 groups x 4 staggered by one barrier, k-tile = 4 phases of 16 MFMAs,
// operands in 16-KiB units, four units in flight behind counted waits), with the two things that kernel leaves on the table:
//
//   1. ONE BLOCK PER CU WALKS MANY OUTPUT TILES and the operand stream never drains: the units of the NEXT output tile are staged
//      during the last k-tile and a half of the current one (unit numbering simply continues across the tile boundary), so there is no
//      64-KiB prologue fill per tile and no idle DMA queue under the epilogue.
//   2. THE EPILOGUE LIVES IN REGISTERS AND IS FOLDED INTO THE MAIN LOOP.  No LDS patch: bias / head RMSNorm / activation / 16-bit
//      packing act on the MFMA accumulator layout directly; `v_permlane16_swap_b32` pairs two 16-column accumulator blocks so a lane
//      owns 8 consecutive 16-bit columns and stores 16 bytes.  The work is cut into six SLOTS placed where the accumulators they
//      touch are final and not yet overwritten:
//          S0 = last k-tile, phase 2     (rows 0..63 of the wave are final since phase 1)       head-norm statistics of rows 0..63
//          S1 = last k-tile, phase 3                                                              quadrant (A0,B0): finish + store
//          S2 = NEXT tile's k-tile 0, phase 0   (before its first MFMA overwrites (A0,B0))       head-norm statistics of rows 64..127
//          S3 = ... phase 1                                                                       quadrant (A0,B1): finish + store
//          S4 = ... phase 2                                                                       quadrant (A1,B1)
//          S5 = ... phase 3                                                                       quadrant (A1,B0)
//      A slot runs in the MEMORY half of its phase, i.e. while the other wave of the SIMD (the other group) is in its 16-MFMA cluster:
//      epilogue VALU overlaps MFMA issue, and the 16 stores per wave and tile trickle out over six phases instead of arriving as a
//      256-store burst per CU at the end.  The first MFMA of the next tile into a quadrant takes a zero C operand.
//      Stores are buffer stores (hardware range check: rows >= M and tiles past the end are dropped without a branch), so every wave
//      issues the SAME number of vector-memory instructions per slot — which the counted waits below rely on.
//
// Counted waits.  vmcnt retires in order (loads, LDS-DMA and stores share the counter on gfx950).  Phase P stages unit P+6 (2 pieces
// per wave) and must know unit P+2 has landed: the ops younger than unit P+2 are the 8 pieces of units P+3..P+6 plus whatever ELSE was
// issued after unit P+2 was staged (in phase P-4, after that phase's slot):  4 stores per store slot (S1, S3, S4, S5) and one parameter
// piece per tile (the next tile's bias row, staged right before its first unit in phase 2 of k-tile nk-2).  With F = the next tile's
// phase 0 the store slots are phases F-1, F+1, F+2, F+3, which gives the table in wait_count().  A count that is too SMALL only waits
// longer; one that is too large would be a race — every entry is derived in the comment next to it.
//
// Epilogue kinds (compile time; anything else stays on gemm_pp256d_kernel):
//   KV   cmap 3: fused condition K|V projection of RDT (bias; K half: per-head RMSNorm + K tiles; V half: Vt tiles — the V-half waves
//        run their MFMAs with the operands SWAPPED, so a lane owns 4 consecutive keys of one d row, which is the Vt tile's order)
//   P16  bias (+ GELU erf / tanh) -> row-major 16-bit C  (ViT qkv / fc1, adaptor first layers)
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_kernels.h"
#include "vt_prof.h"

extern int g_vt_gm;

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) unsigned uint4_t;
typedef __attribute__((ext_vector_type(2))) unsigned uint2_t;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int BUF_BYTES = (BM + BN) * 128;        // one k-tile: A rows 0..255 then B rows 0..255, 128 B each
constexpr int PAR_OFF = 2 * BUF_BYTES;            // parameter area: [slot 2][wave 8][64 floats bias | 64 floats column scale] of the wave's 64 columns
constexpr int PAR_WAVE = 512;
constexpr int HN_OFF = PAR_OFF + 2 * 8 * PAR_WAVE;   // 64 floats: head-norm gains (k_norm) of the KV kind
constexpr int SMEM_BYTES = HN_OFF + 256;

enum { KIND_KV = 0, KIND_P16 = 1, KIND_R32 = 2 };
enum { MODE_STEADY = 0, MODE_FIRST = 1, MODE_SECOND = 2, MODE_SWITCH = 3, MODE_LAST = 4 };

// the epilogue's store; bench builds can turn it into a register sink (VLATOUCH_PT_ABL & 8: what do the stores themselves cost?)
__device__ __forceinline__ void pt_store(const uint4_t d, const __amdgpu_buffer_rsrc_t rc, const int off, const int abl) {
#ifdef VLATOUCH_BENCH_BUILD
  if (abl & 8) { asm volatile("" :: "v"(d), "v"(off)); return; }
#endif
#ifdef VLATOUCH_BENCH_BUILD      // A/B of the store's cache policy bits (VLATOUCH_PT_ABL bits 4..6 -> aux: 1 = sc0, 2 = nt, 3 = sc0 nt)
  if ((abl >> 4) & 1) { __builtin_amdgcn_raw_buffer_store_b128(d, rc, off, 0, 2); return; }
  if ((abl >> 5) & 1) { __builtin_amdgcn_raw_buffer_store_b128(d, rc, off, 0, 3); return; }
  if ((abl >> 6) & 1) { __builtin_amdgcn_raw_buffer_store_b128(d, rc, off, 0, 1); return; }
#endif
  __builtin_amdgcn_raw_buffer_store_b128(d, rc, off, 0, 0);
}

// compile-time loop: indices are constants by construction (an epilogue slot that the unroller gives up on would index the accumulators
// dynamically and push all 128 of them to scratch)
template <int I0, int I1, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) { f(std::integral_constant<int, I0>{}); static_for<I0 + 1, I1>(f); }
}

// a value the optimiser must treat as new at this point: derived addresses cannot be hoisted out of the tile walk as loop invariants
// (hipcc hoisted ~50 per-lane store offsets of the epilogue slots to the kernel's entry and spilled them)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// the lane id, recomputed where it is needed (2 VALU ops, no live register across the main loop: a lane constant kept for the slots was the one
// value hipcc still spilled, and every reload of it cost an `s_waitcnt vmcnt(0)`, i.e. a drained operand stream)
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// parameter reads from LDS as inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of every ordinary ds_read it cannot prove disjoint from a pending LDS-DMA
// (the parameter pieces land in this same array), which would drain the operand stream once per slot.  The rows read here were staged a whole k-tile
// (>= 4 counted waits + barriers) earlier.  lds_wait names the registers it retires so that no use can be scheduled above it.
template <int OFF> __device__ __forceinline__ float4_t lds_ld128(unsigned addr) {
  float4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF> __device__ __forceinline__ float lds_ld32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void lds_wait(float4_t& a, float4_t& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lds_wait(float4_t& a, float4_t& b, float4_t& c, float4_t& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void lds_wait(float& a, float& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }

// sum over the four 16-lane rows of a wave (lanes that differ in bits 4 and 5), result in every lane, on the VALU only: v_permlane16_swap pairs rows
// (0,1) and (2,3), v_permlane32_swap the two halves — a __shfl_xor (ds_bpermute) costs an LDS round trip per step, and a slot runs these on its
// critical path
__device__ __forceinline__ float sum_rows4(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
  const unsigned x = __builtin_bit_cast(unsigned, w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

#ifdef VLATOUCH_DRAIN_WAITS      // debug build (tools/drain_waits_check.sh): every counted wait drains the whole queue — results must not change by a bit
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif

// ops allowed to stay in flight after phase `ph` of a k-tile of kind MODE has staged its unit (see the header).  NST = 4 stores per store slot.
//   STEADY                       8 pieces
//   SWITCH (k-tile nk-2)         ph 2, 3: + the parameter piece issued at the head of ph 2 (younger than the target units P+2 <= L-2+3+2 ... staged before it)
//   LAST   (k-tile nk-1)         ph 0, 1: + that piece (target unit staged in SWITCH ph 0 / 1, before it); ph 2: 8 (target staged in SWITCH ph 2
//                                AFTER the piece); ph 3 = phase F-1: + S1's 4 stores (issued this phase, before the stage)
//   FIRST  (next tile, k-tile 0) phases F..F+3: store slots among P-3..P: {F-1} / {F-1,F+1} / {F-1,F+1,F+2} / {F+1,F+2,F+3} -> 12 / 16 / 20 / 20
//   SECOND (k-tile 1)            phases F+4..F+7: {F+1,F+2,F+3} / {F+2,F+3} / {F+3} / {} -> 20 / 16 / 12 / 8
//   R32 kind: TWO parameter pieces per tile (bias, column scale) and no store slots; its 32 stores per wave are issued in one flat block between the
//   last phase of a tile and phase F of the next: every FIRST phase's target unit (F+2 .. F+5) was staged BEFORE them -> 8 + 32 = 40; SECOND: 8.
__host__ __device__ constexpr int wait_count(int kind, int mode, int ph) {
  return kind == KIND_R32
       ? (mode == MODE_SWITCH ? (ph < 2 ? 8 : 10) : mode == MODE_LAST ? (ph < 2 ? 10 : 8) : mode == MODE_FIRST ? 40 : 8)
       : (mode == MODE_SWITCH ? (ph < 2 ? 8 : 9)
        : mode == MODE_LAST   ? (ph < 2 ? 9 : (ph == 2 ? 8 : 12))
        : mode == MODE_FIRST  ? (ph == 0 ? 12 : (ph == 1 ? 16 : 20))
        : mode == MODE_SECOND ? (ph == 0 ? 20 : (ph == 1 ? 16 : (ph == 2 ? 12 : 8)))
        : 8);
}

template <typename T16> __device__ __forceinline__ unsigned pack16(float a, float b);
template <> __device__ __forceinline__ unsigned pack16<bf16_t>(float a, float b) { return pk_bf16(a, b); }
template <> __device__ __forceinline__ unsigned pack16<half_t>(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  return __builtin_bit_cast(unsigned, (h2_t){(_Float16)a, (_Float16)b});
}

__device__ __forceinline__ void mma16z(float4_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), (float4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
__device__ __forceinline__ void mma16z(float4_t& acc, const Frag<half_t>& a, const Frag<half_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a.v), __builtin_bit_cast(f16x8_t, b.v), (float4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

// the output tile a block works on, and everything the epilogue of that tile needs (all wave-uniform)
struct TileId { int m0, n0; };

// SWAP (KV kind only): this block walks V-half tiles (columns >= N/2) and runs its MFMAs with the operands exchanged.  `tiles_n`, `total_tiles`
// count the tiles of ONE role (KV: one half of the columns); `nroles` = 2 for the KV kind: blocks with ((blockIdx.x >> 3) & 1) == 1 are the V role.
template <typename T16, int KIND, int ACT, bool SWAP>
__device__ __forceinline__ void pt_body(const VtGemmParams& p, char* smem, const int tiles_n, const int tiles_m, const int total_tiles, const int GM, const int nroles,
                                        const int abl, const bool rev = false) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;          // group (row half), column quarter
  const int g = lane >> 4, l15 = lane & 15;
  const int nk = p.K / BK;

  // ---- persistent tile walk: block b = (XCD x = b & 7, slot s = b >> 3) takes entries s, s + S, s + 2S, ... of XCD x's contiguous band of the tile
  //      order (the order itself — super-rows of GM m-tiles, n-major inside — is gemm_pp256d_kernel's), so the S blocks of an XCD work on neighbouring tiles
  const int G = gridDim.x / nroles;                                  // blocks of this role
  // this block's index among them: blockIdx.x with the role bit (bit rb >= 3, the XCD bits stay) squeezed out
  const int rb = (abl >> 8) & 15;
  const int bx = nroles == 2 ? (int)(((blockIdx.x >> (rb + 1)) << rb) | (blockIdx.x & ((1u << rb) - 1))) : (int)blockIdx.x;
  const int n_base = SWAP ? (p.N >> 1) : 0;
  const bool banded = (G & 7) == 0;
  const int S = banded ? (G >> 3) : G;
  const int band = banded ? (total_tiles + 7) / 8 : total_tiles;
  const int band0 = banded ? (bx & 7) * band : 0;
  const int bandn = min(band, total_tiles - band0);                 // may be <= 0 for the last XCDs of a tiny grid
  int idx = banded ? (bx >> 3) : bx;
  if (rev) idx = S - 1 - idx;       // second phase of a two-phase walk: the blocks that had one tile more in the first phase get one less here
  if (idx >= bandn) return;
#ifdef VLATOUCH_BENCH_BUILD      // 2 = only the K-role blocks run, 4 = only the V-role blocks (KV kind: how long does each role take alone?)
  if ((abl & 2) && SWAP) return;
  if ((abl & 4) && !SWAP && KIND == KIND_KV) return;
#endif
  auto decode = [&](int id) __attribute__((always_inline)) -> TileId {
    const int sr = id / (GM * tiles_n);
    const int gmr = min(GM, tiles_m - sr * GM);
    const int r_in = id - sr * GM * tiles_n;
    const int tn = r_in / gmr, tm = sr * GM + (r_in - tn * gmr);
    return TileId{tm * BM, n_base + tn * BN};
  };

  // ---- staging context (of the tile whose units are being issued): buffer resources of its A / W row blocks + per-lane offsets.
  //      Pieces as in gemm_pp256d_kernel: a unit = 128 tile rows = 16 pieces of 8 rows, this wave issues pieces 2*wave, 2*wave+1.
  __amdgpu_buffer_rsrc_t rsA, rsW;
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
      dst[u][e] = (isA ? 0 : BM * 128) + r8 * 128;
    }
  const float* bias = p.bias;
  auto set_stage_ctx = [&](const TileId t) __attribute__((always_inline)) {
#ifdef VLATOUCH_BENCH_BUILD      // timing only (VLATOUCH_PT_ABL & 128, garbage results): every tile stages the A rows of m-tile (tm & 1) — the A panel becomes
    // L2-resident (2 MB per XCD), the fabric reads of the launch drop to the W stream: does the time follow the traffic?
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (long)((abl & 128) ? (t.m0 & BM) : t.m0) * p.lda;
#else
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (long)t.m0 * p.lda;
#endif
    const uint16_t* W = reinterpret_cast<const uint16_t*>(p.W) + (long)t.n0 * p.ldw;
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
    const int ln = fresh_lane();
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = 2 * wave + e;
        const bool isA = (u == 0 || u == 3);
        const int sel = (u == 2 || u == 3) ? 1 : 0;
        const int r8 = isA ? ((q >> 3) * 128 + sel * 64 + (q & 7) * 8) : ((q >> 2) * 64 + sel * 32 + (q & 3) * 8);
        const int r = r8 + (ln >> 3);
        const int c = (ln & 7) ^ ((r >> 1) & 7);
        src[u][e] = isA ? (int)(((long)(min(t.m0 + r, p.M - 1) - t.m0) * p.lda + c * 8) * 2) : (int)(((long)(min(t.n0 + r, p.N - 1) - t.n0) * p.ldw + c * 8) * 2);
      }
  };
  auto stage = [&](int u, int buf, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds((u == 0 || u == 3) ? rsA : rsW, (lds_void*)(smem + buf * BUF_BYTES + dst[u][e]), 16, src[u][e], kt * (BK * 2), 0, 0);
  };
  // the parameter piece of a tile: the 64 bias values of this wave's columns -> LDS slot `slot` (one 4-byte-per-lane DMA; columns >= N clamp)
  auto stage_params = [&](const TileId t, int slot) __attribute__((always_inline)) {
    const int n = min(t.n0 + wn * 64 + lane, p.N - 1);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(smem + PAR_OFF + (slot * 8 + wave) * PAR_WAVE), 4, n * 4, 0, 0, 0);
    if constexpr (KIND == KIND_R32) {
      const __amdgpu_buffer_rsrc_t rcs = __builtin_amdgcn_make_buffer_rsrc((void*)p.colscale, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rcs, (lds_void*)(smem + PAR_OFF + (slot * 8 + wave) * PAR_WAVE + 256), 4, n * 4, 0, 0, 0);
    }
  };

  float4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const unsigned smem_base = (unsigned)(size_t)(lds_void*)smem;
  // ---- epilogue state: the tile being finished (`et`), its parameter slot, whether this wave ran it with swapped operands, the row statistics
  TileId et{0, 0};
  int eslot = 0;
  constexpr bool eswap = SWAP;
  float rstd[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) rstd[h][j] = 0.f;


  // ================================================================ epilogue slots (all on registers; see the header)
  // statistics of the wave's rows JB*16 .. JB*16+63 (K half of the KV kind): x = acc + bias kept in place, rstd per row
  auto epi_stats = [&](auto jb_tag, float (&rs)[4]) {
    constexpr int JB = decltype(jb_tag)::value;
    if constexpr (KIND == KIND_KV) {
      if constexpr (eswap) return;
      const unsigned pa = smem_base + PAR_OFF + (eslot * 8 + wave) * PAR_WAVE + (fresh_lane() >> 4) * 16;
      float4_t b4[4];
      b4[0] = lds_ld128<0>(pa); b4[1] = lds_ld128<64>(pa); b4[2] = lds_ld128<128>(pa); b4[3] = lds_ld128<192>(pa);
      lds_wait(b4[0], b4[1], b4[2], b4[3]);
      static_for<0, 4>([&](auto jj_) __attribute__((always_inline)) {
        constexpr int jj = decltype(jj_)::value, j = JB + jj;
        float q = 0.f, sm = 0.f;
        static_for<0, 4>([&](auto i_) __attribute__((always_inline)) {
          constexpr int i = decltype(i_)::value;
          float4_t x = acc[i][j];
          x += b4[i];
          acc[i][j] = x;
          q = fmaf(x[0], x[0], q); q = fmaf(x[1], x[1], q); q = fmaf(x[2], x[2], q); q = fmaf(x[3], x[3], q);
          if (p.hn_mode == 2) sm += (x[0] + x[1]) + (x[2] + x[3]);
        });
        q = sum_rows4(q);
        float var;
        if (p.hn_mode == 2) {
          sm = sum_rows4(sm);
          const float mean = sm * (1.f / 64.f);
          var = (q - 64.f * mean * mean) * (1.f / 63.f);
        } else var = q * (1.f / 64.f);
        rs[jj] = rsqrtf(var + p.hn_eps);
      });
    }
  };
  // finish + store one quadrant: rows JB*16 .. +63 (4 j), column blocks IB, IB+1.  Always exactly 4 buffer stores of 16 bytes per lane.
  auto epi_store = [&](auto jb_tag, auto ib_tag, const float (&rs)[4]) {
    constexpr int JB = decltype(jb_tag)::value, IB = decltype(ib_tag)::value;
    const int mrow0 = et.m0 + wm * 128, ncol0 = et.n0 + wn * 64;
    const unsigned pw = smem_base + PAR_OFF + (eslot * 8 + wave) * PAR_WAVE;
    const int fl = fresh_lane(), fg = fl >> 4, fl15 = fl & 15;
    const int flbK = fl15 * 128 + ((fg & 1) * 16 + (fg >> 1) * 8) * 2;      // K tile / row-major: row l15, the lane's 8 columns after the permlane swap
    if constexpr (KIND == KIND_KV) {
      // tile pair (K tile, Vt tile) of head h and row block t: 16 KiB at ((h * T + t) * 2) * 4096 elements; this wave touches t0 + (j >> 2)
      const int half = p.N >> 1;
      const int h = (eswap ? ncol0 - half : ncol0) >> 6;
      const int t0 = mrow0 >> 6;
      const long tiles_left = (long)p.cmap_T - t0;                  // <= 0: the wave's rows are all past the end (edge m-tile)
      char* base = reinterpret_cast<char*>(p.C) + ((long)h * p.cmap_T + t0) * 16384;
      const int nrec = tiles_left >= 2 ? 32768 : (tiles_left == 1 ? 16384 : 0);
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nrec, 0x00020000);
      if constexpr (!eswap) {
        // K half: lane = row j*16 + l15, columns i*16 + g*4 + r.  o = x * (rstd * gain)  (x already holds acc + bias)
        const unsigned ha = smem_base + HN_OFF + fg * 16;
        float4_t g4[2];
        g4[0] = lds_ld128<IB * 64>(ha); g4[1] = lds_ld128<IB * 64 + 64>(ha);
        lds_wait(g4[0], g4[1]);
        const int lb = flbK;
        static_for<0, 4>([&](auto jj_) __attribute__((always_inline)) {
          constexpr int jj = decltype(jj_)::value, j = JB + jj;
          unsigned d[2][2];
          static_for<0, 2>([&](auto ii_) __attribute__((always_inline)) {
            constexpr int ii = decltype(ii_)::value;
            const float4_t x = acc[IB + ii][j];
            const float r_ = rs[jj];
            d[ii][0] = pack16<T16>(x[0] * (r_ * g4[ii][0]), x[1] * (r_ * g4[ii][1]));
            d[ii][1] = pack16<T16>(x[2] * (r_ * g4[ii][2]), x[3] * (r_ * g4[ii][3]));
          });
          const auto s0 = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
          const int m = mrow0 + j * 16 + fl15;
          const int off = lb + ((j >> 2) * 16384 + (j & 3) * 2048 + IB * 32);
          pt_store((uint4_t){s0[0], s1[0], s0[1], s1[1]}, rc, m < p.M ? off : 0x7ffffff0, abl);
        });
      } else {
        // V half (swapped operands): lane = d row i*16 + l15, keys j*16 + g*4 + r; Vt position of key kk: vt_kpos -> (j&2)*16 + g*8 + (j&1)*4 + r,
        // so the pair (j even, j odd) is 8 consecutive positions = 16 bytes
        const bool full = mrow0 + 128 <= p.M;
        const int lb = fl15 * 128 + fg * 16;      // Vt tile: d row l15, 8 key positions of lane group g
        const int mg = mrow0 + fg * 4;
        const unsigned va = pw + fl15 * 4;
        float bvs[2];
        bvs[0] = lds_ld32<IB * 64>(va); bvs[1] = lds_ld32<IB * 64 + 64>(va);
        lds_wait(bvs[0], bvs[1]);
        static_for<0, 2>([&](auto ii_) __attribute__((always_inline)) {
          constexpr int i = IB + decltype(ii_)::value;
          const float bv = bvs[decltype(ii_)::value];
          static_for<0, 2>([&](auto jp_) __attribute__((always_inline)) {
            constexpr int j = JB + 2 * decltype(jp_)::value;
            float4_t lo = acc[i][j], hi = acc[i][j + 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) { lo[r] += bv; hi[r] += bv; }
            if (!full) {
              const int mlo = mg + j * 16, mhi = mlo + 16;
#pragma unroll
              for (int r = 0; r < 4; ++r) { lo[r] = mlo + r < p.M ? lo[r] : 0.f; hi[r] = mhi + r < p.M ? hi[r] : 0.f; }
            }
            const int off = lb + ((j >> 2) * 16384 + 8192 + i * 2048 + (j & 2) * 32);
            // a row block past the last tile of the stream (the second of the two tiles when only one is left) must not be written: the range
            // check covers voffset only up to num_records, which is per WAVE here -> send it out of range by hand
            pt_store((uint4_t){pack16<T16>(lo[0], lo[1]), pack16<T16>(lo[2], lo[3]), pack16<T16>(hi[0], hi[1]), pack16<T16>(hi[2], hi[3])},
                     rc, (mg + j * 16) < p.cmap_T * 64 ? off : 0x7ffffff0, abl);
          });
        });
      }
    } else {
      // row-major 16-bit C: lane = row j*16 + l15, columns i*16 + g*4 + r; o = act(acc + bias)
      const long rows_left = (long)p.M - mrow0;
      const int cols_ok = ncol0 < p.N;                                           // N % 64 == 0: a wave's columns are all in or all out
      char* base = reinterpret_cast<char*>(p.C) + ((long)mrow0 * p.ldc + ncol0) * 2;
      const long nrec = (rows_left > 0 && cols_ok) ? (min(rows_left, 128L) - 1) * p.ldc * 2 + 128 : 0;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nrec, 0x00020000);
      const unsigned pa = pw + fg * 16;
      float4_t b4[2];
      b4[0] = lds_ld128<IB * 64>(pa); b4[1] = lds_ld128<IB * 64 + 64>(pa);
      lds_wait(b4[0], b4[1]);
      const int rowb = (int)p.ldc * 2;
      const int lb = fl15 * rowb + (flbK & 127);
      static_for<0, 4>([&](auto jj_) __attribute__((always_inline)) {
        constexpr int j = JB + decltype(jj_)::value;
        unsigned d[2][2];
        static_for<0, 2>([&](auto ii_) __attribute__((always_inline)) {
          constexpr int ii = decltype(ii_)::value;
          const float4_t x = acc[IB + ii][j];
          float o0 = x[0] + b4[ii][0], o1 = x[1] + b4[ii][1], o2 = x[2] + b4[ii][2], o3 = x[3] + b4[ii][3];
          if constexpr (ACT != VT_ACT_NONE) { o0 = act_apply(o0, ACT); o1 = act_apply(o1, ACT); o2 = act_apply(o2, ACT); o3 = act_apply(o3, ACT); }
          d[ii][0] = pack16<T16>(o0, o1);
          d[ii][1] = pack16<T16>(o2, o3);
        });
        const auto s0 = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false);
        const int off = lb + j * 16 * rowb + IB * 32;                            // rows >= M land past num_records: dropped by the range check
        pt_store((uint4_t){s0[0], s1[0], s0[1], s1[1]}, rc, off, abl);
      });
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I2 = std::integral_constant<int, 2>;
  using I4 = std::integral_constant<int, 4>;
  // R32 kind: C (fp32) = residual + colscale * (acc + bias), the whole 128 x 64 sub-tile of the wave in one flat block at the end of the tile, two
  // quadrants at a time: 16 residual loads (64 registers: the fragments are dead here) in flight, then their arithmetic and 16 stores.  The residual
  // loads are ordinary VGPR loads, so hipcc drains the wave's queue at their first use — the next tile's first six units are in flight by then and
  // are needed next anyway.  (Folded into the next tile's k-loop like the 16-bit kinds, the loads would need 32 more registers than the wave
  // has, or a counted wait that keeps only ONE operand unit in flight for four phases.)
  auto epi_flat = [&]() __attribute__((always_inline)) {
    if constexpr (KIND == KIND_R32) {
      const int mrow0 = et.m0 + wm * 128, ncol0 = et.n0 + wn * 64;
      const int fl = fresh_lane(), fg = fl >> 4, fl15 = fl & 15;
      const unsigned pa = smem_base + PAR_OFF + (eslot * 8 + wave) * PAR_WAVE + fg * 16;
      float4_t b4[4], c4[4];
      b4[0] = lds_ld128<0>(pa); b4[1] = lds_ld128<64>(pa); b4[2] = lds_ld128<128>(pa); b4[3] = lds_ld128<192>(pa);
      c4[0] = lds_ld128<256>(pa); c4[1] = lds_ld128<320>(pa); c4[2] = lds_ld128<384>(pa); c4[3] = lds_ld128<448>(pa);
      lds_wait(b4[0], b4[1], b4[2], b4[3]);
      lds_wait(c4[0], c4[1], c4[2], c4[3]);
      const long rows_left = (long)p.M - mrow0;
      const bool ok = rows_left > 0 && ncol0 < p.N;
      const long nrows = rows_left < 128 ? rows_left : 128;
      char* cbase = reinterpret_cast<char*>(p.C) + ((long)mrow0 * p.ldc + ncol0) * 4;
      const char* rbase = reinterpret_cast<const char*>(p.residual) + ((long)mrow0 * p.ldr + ncol0) * 4;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)cbase, 0, ok ? (int)((nrows - 1) * p.ldc * 4 + 256) : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)rbase, 0, ok ? (int)((nrows - 1) * p.ldr * 4 + 256) : 0, 0x00020000);
      const int crow = (int)p.ldc * 4, rrow = (int)p.ldr * 4;
      const int clane = fl15 * crow + fg * 16, rlane = fl15 * rrow + fg * 16;
      static_for<0, 2>([&](auto h_) __attribute__((always_inline)) {          // rows 0..63, then rows 64..127 of the wave
        constexpr int JB = decltype(h_)::value * 4;
        uint4_t rv[4][4];
        static_for<0, 4>([&](auto jj_) __attribute__((always_inline)) {
          static_for<0, 4>([&](auto i_) __attribute__((always_inline)) {
            constexpr int jj = decltype(jj_)::value, i = decltype(i_)::value, j = JB + jj;
            rv[jj][i] = __builtin_amdgcn_raw_buffer_load_b128(rr, rlane + j * 16 * rrow + i * 64, 0, 0);
          });
        });
        static_for<0, 4>([&](auto jj_) __attribute__((always_inline)) {
          static_for<0, 4>([&](auto i_) __attribute__((always_inline)) {
            constexpr int jj = decltype(jj_)::value, i = decltype(i_)::value, j = JB + jj;
            float4_t x = acc[i][j];
            const float4_t r = __builtin_bit_cast(float4_t, rv[jj][i]);
            x = (x + b4[i]) * c4[i] + r;
            pt_store(__builtin_bit_cast(uint4_t, x), rc, clane + j * 16 * crow + i * 64, abl);
          });
        });
      });
    }
  };
  auto slot_run = [&](auto s_tag) __attribute__((always_inline)) {
    constexpr int SL = decltype(s_tag)::value;
    if constexpr (KIND == KIND_R32) return;
#ifdef VLATOUCH_BENCH_BUILD      // timing-only ablations (tools/gemm_bench_pt.py --abl; garbage results): 1 = no epilogue slots at all
    if (abl & 1) return;
#endif
    if constexpr (SL == 0) epi_stats(I0{}, rstd[0]);
    else if constexpr (SL == 1) epi_store(I0{}, I0{}, rstd[0]);       // (A0, B0)
    else if constexpr (SL == 2) epi_stats(I4{}, rstd[1]);
    else if constexpr (SL == 3) epi_store(I0{}, I2{}, rstd[0]);       // (A0, B1)
    else if constexpr (SL == 4) epi_store(I4{}, I2{}, rstd[1]);       // (A1, B1)
    else epi_store(I4{}, I0{}, rstd[1]);                                // (A1, B0)
  };

  // ================================================================ prologue
  TileId cur = decode(band0 + idx);
  set_stage_ctx(cur);
  if constexpr (KIND == KIND_KV) {
    if (wave == 0) reinterpret_cast<float*>(smem + HN_OFF)[lane] = p.hn_w0 ? p.hn_w0[lane] : 1.f;
  }
  stage_params(cur, 0);
#pragma unroll
  for (int n = 0; n < 6; ++n) stage(n & 3, (n >> 2) & 1, n >> 2);          // nk >= 4: units 0..5 exist
  wait_vm<8>();                                                             // the parameter piece(s) and units 0, 1 landed
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // wave 0's gain row (a __syncthreads() here would drain the DMA queue)
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();       // stagger group 1 by one barrier (group 0 pays it back after the loop)

  const int arow = wm * 128 + l15, brow = wn * 64 + l15;
  int par = 0;                                     // buffer parity of this tile's k-tile 0 (flips per tile when nk is odd)
  int tseq = 0;
  TileId nxt = cur;

  // one k-tile.  MODE selects the wait counts, which epilogue slots run and (SWITCH) where the staging context moves to the next tile.
  auto ktile = [&](const int kt, auto mode_tag) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    const int cur_b = (kt + par) & 1;
    const char* As = smem + cur_b * BUF_BYTES;
    const char* Bs = As + BM * 128;
    Frag<T16> a0[4][2], a1[4][2], b0[2][2], b1[2][2];
    // head of a phase, BEFORE its fragment reads (fewest fragments live: the slot's temporaries fit beside the 128 accumulators): the epilogue slot
    // of this phase and, in SWITCH phase 2, the move of the staging context to the next tile
    auto head = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int ph = decltype(ph_tag)::value;
#ifdef VLATOUCH_PT_SLOT_HEAD      // A/B: the first placement of the slots (head of the phase, before its fragment reads)
      if constexpr (MODE == MODE_LAST && ph == 2) slot_run(std::integral_constant<int, 0>{});
      if constexpr (MODE == MODE_LAST && ph == 3) slot_run(std::integral_constant<int, 1>{});
      if constexpr (MODE == MODE_FIRST) slot_run(std::integral_constant<int, 2 + ph>{});
#endif
      if constexpr (MODE == MODE_SWITCH && ph == 2) {           // from here on the stream belongs to the next tile (or, past the last one, re-reads this one)
        set_stage_ctx(nxt);
        stage_params(nxt, (tseq + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // tail of a phase = right behind its 16 MFMAs, before the barrier that ends the compute half: the epilogue slot that the FIRST placement ran at the
    // head of the NEXT phase.  Same position in the wave's vector-memory queue (after this phase's unit, before the next one: the wait table is
    // unchanged) and the same live fragments, but the phase's own fragment reads were issued long ago and the slot's VALU starts while the cluster's last
    // MFMAs still execute.  At the head the slot sat between the other group's MFMA cluster and this group's fragment reads, so every slot phase
    // cost slot + exposed LDS latency, twice (once per group): measured 8 % of the K|V projection for the arithmetic alone (tools/pt_abl.sh).
    auto tail = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int ph = decltype(ph_tag)::value;
#ifndef VLATOUCH_PT_SLOT_HEAD
      if constexpr (MODE == MODE_LAST && ph == 1) slot_run(std::integral_constant<int, 0>{});
      if constexpr (MODE == MODE_LAST && ph == 2) slot_run(std::integral_constant<int, 1>{});
      if constexpr (MODE == MODE_LAST && ph == 3) slot_run(std::integral_constant<int, 2>{});
      if constexpr (MODE == MODE_FIRST && ph < 3) slot_run(std::integral_constant<int, 3 + ph>{});
#endif
    };
    auto mem_end = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int ph = decltype(ph_tag)::value;
      constexpr int U = (ph + 2) & 3, DT = (ph + 6) >> 2;      // unit P+6 = (k-tile kt + DT, U)
      int ks = kt + DT;                                         // k-tile index inside the tile the unit belongs to
      if constexpr (MODE == MODE_SWITCH) { if (ph >= 2) ks = 0; }
      if constexpr (MODE == MODE_LAST) ks = ph < 2 ? 0 : 1;
      stage(U, (kt + DT + par) & 1, ks);
      wait_vm<wait_count(KIND, MODE, ph)>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    // 16 MFMAs of one quadrant; FIRST: the ks = 0 MFMAs start from a zero C (the quadrant's previous contents were stored by its slot)
#define VT_PT_MMA(AF, BF, ah, bh)                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                         \
    static_for<0, 16>([&](auto n_) __attribute__((always_inline)) {                                                        \
      constexpr int n = decltype(n_)::value, ks = n >> 3, i = (n >> 2) & 1, j = n & 3;                                      \
      float4_t& c_ = acc[(bh) * 2 + i][(ah) * 4 + j];                                                                       \
      if constexpr (!SWAP) {                                                                                               \
        if constexpr (MODE == MODE_FIRST && ks == 0) mma16z(c_, BF[i][ks], AF[j][ks]); else mma16(c_, BF[i][ks], AF[j][ks]); \
      } else {                                                                                                             \
        if constexpr (MODE == MODE_FIRST && ks == 0) mma16z(c_, AF[j][ks], BF[i][ks]); else mma16(c_, AF[j][ks], BF[i][ks]); \
      }                                                                                                                    \
    });                                                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                                         \
    tail(std::integral_constant<int, (ah) == 0 ? (bh) : 3 - (bh)>{});      /* (A0,B0) (A0,B1) (A1,B1) (A1,B0) = phases 0 1 2 3 */ \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    __builtin_amdgcn_s_barrier();                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
    // phase 0: quadrant (A0, B0)
    head(std::integral_constant<int, 0>{});
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(b0[i][ks], Bs, brow + i * 16, ks * 4 + g);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(a0[j][ks], As, arow + j * 16, ks * 4 + g);
    mem_end(std::integral_constant<int, 0>{});
    VT_PT_MMA(a0, b0, 0, 0)
    // phase 1: quadrant (A0, B1)
    head(std::integral_constant<int, 1>{});
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(b1[i][ks], Bs, brow + 32 + i * 16, ks * 4 + g);
    mem_end(std::integral_constant<int, 1>{});
    VT_PT_MMA(a0, b1, 0, 1)
    // phase 2: quadrant (A1, B1)
    head(std::integral_constant<int, 2>{});
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lds_frag(a1[j][ks], As, arow + 64 + j * 16, ks * 4 + g);
    mem_end(std::integral_constant<int, 2>{});
    VT_PT_MMA(a1, b1, 1, 1)
    // phase 3: quadrant (A1, B0), no reads
    head(std::integral_constant<int, 3>{});
    mem_end(std::integral_constant<int, 3>{});
    VT_PT_MMA(a1, b0, 1, 0)
#undef VT_PT_MMA
  };

  // ================================================================ the tile walk
  for (;;) {
    const int nidx = idx + S;
    const bool has_next = nidx < bandn;
    nxt = has_next ? decode(band0 + nidx) : cur;     // past the last tile the stream re-reads this tile's first units (never consumed): the code stays uniform
    int kt = 0;
    if (tseq > 0) {
      ktile(0, std::integral_constant<int, MODE_FIRST>{});
      ktile(1, std::integral_constant<int, MODE_SECOND>{});
      kt = 2;
    }
    for (; kt < nk - 2; ++kt) ktile(kt, std::integral_constant<int, MODE_STEADY>{});
    ktile(nk - 2, std::integral_constant<int, MODE_SWITCH>{});
    et = cur; eslot = tseq & 1;                      // slots S0 .. S5 finish THIS tile (S2 .. S5 inside the next tile's first k-tile)
    ktile(nk - 1, std::integral_constant<int, MODE_LAST>{});
    epi_flat();                                      // R32 kind only
    if (!has_next) break;
    cur = nxt; idx = nidx; ++tseq;
    par ^= (nk & 1);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();       // pay back the stagger
  // the last tile's remaining slots, back to back; then drain the (never consumed) trailing units before the wave ends
#ifdef VLATOUCH_PT_SLOT_HEAD
  slot_run(std::integral_constant<int, 2>{});
#endif
  slot_run(std::integral_constant<int, 3>{});
  slot_run(std::integral_constant<int, 4>{});
  slot_run(std::integral_constant<int, 5>{});
  wait_vm<0>();
}

// KV kind: the K-half and the V-half tiles are walked by DIFFERENT blocks (roles alternate in groups of 8 blocks = one block per XCD), because the
// V half runs its MFMAs with exchanged operands and that choice has to be compile time for the register allocator: the same 16 (of 32) blocks of an
// XCD walk the same band of m-tiles in both roles, so the A panel they stream is shared in that XCD's L2.
template <typename T16, int KIND, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_pt_kernel(const VtGemmParams p, const int tiles_n, const int tiles_m, const int total_tiles, const int GM, const int abl) {
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  if constexpr (KIND == KIND_KV) {
    if (abl & (1 << 12)) {        // A/B (VLATOUCH_PT_KV_SPLIT=1): the first form — K-half and V-half tiles walked by DIFFERENT blocks (alternate groups of 8)
      if ((blockIdx.x >> ((abl >> 8) & 15)) & 1) pt_body<T16, KIND, ACT, true>(p, smem, tiles_n, tiles_m, total_tiles, GM, 2, abl);
      else pt_body<T16, KIND, ACT, false>(p, smem, tiles_n, tiles_m, total_tiles, GM, 2, abl);
    } else {
      // every block walks its share of the K-half tiles, then of the V-half tiles (the operand swap is compile time, so the two walks are two copies
      // of the loop nest): all blocks of an XCD are in the same half at the same time and at the same pace, so the A panel and the W band they share stay
      // shared, and no CU idles while the slower half finishes.  The second walk takes the slots in reverse, which evens out the odd tile.
      pt_body<T16, KIND, ACT, false>(p, smem, tiles_n, tiles_m, total_tiles, GM, 1, abl, false);
      __syncthreads();            // (LDS of the first walk is dead: its last parameter reads are behind this barrier)
      pt_body<T16, KIND, ACT, true>(p, smem, tiles_n, tiles_m, total_tiles, GM, 1, abl, true);
    }
  } else {
    pt_body<T16, KIND, ACT, false>(p, smem, tiles_n, tiles_m, total_tiles, GM, 1, abl);
  }
}

int g_pt_on = -1;
int g_pt_abl = 0;                // bench builds only (VLATOUCH_PT_ABL): see the #ifdef VLATOUCH_BENCH_BUILD blocks above
int g_pt_cus = 0;

}  // namespace


// epilogue kinds the persistent kernel has (anything else stays on gemm_pp256d_kernel)
static bool pt_kind_ok(const VtGemmParams& p) {
  if (p.groups != 1 || p.K < 4 * BK || !p.bias || (p.N % 64)) return false;
  if (p.lda >= (1 << 21) || p.ldw >= (1 << 21) || p.K >= (1 << 24)) return false;   // 32-bit buffer offsets inside a 256-row block
  if (p.c_dtype == VT_F32)          // R32 kind: fp32 C = residual + colscale * (acc + bias)
    return p.residual && p.colscale && p.cmap == 0 && !p.hn_w0 && !p.hn_w1 && p.act == VT_ACT_NONE && (long)p.ldc * 4 * 128 < (1L << 31) && (long)p.ldr * 4 * 128 < (1L << 31);
  if (p.residual || p.colscale) return false;
  if (p.cmap == 3) return (p.a_dtype == VT_BF16 || p.a_dtype == VT_F16) && (p.N % 512) == 0 && p.hn_c0_end == (p.N >> 1) && !p.hn_w1 && p.act == VT_ACT_NONE && p.hn_w0;
  if (p.cmap != 0 || p.hn_w0 || p.hn_w1) return false;
  if ((long)p.ldc * 2 * 128 >= (1L << 31)) return false;
  return p.act == VT_ACT_NONE || p.act == VT_ACT_GELU_ERF || p.act == VT_ACT_GELU_TANH;
}
static bool pt_enabled() {
  if (g_pt_on < 0) { const char* e = getenv("VLATOUCH_PT"); g_pt_on = e ? atoi(e) : 1; }
  return g_pt_on != 0;
}

void vt_gemm_pt_tune(int value) {
  (void)pt_enabled();                        // the environment default is read before the first explicit setting
  g_pt_on = value != 0;
}

// One round of 160 .. 256 tiles at K >= 512 (DINOv2-B out-projection: 64 x 3 tiles, K = 768): too short for gemm_pp256d_kernel to beat the 128-column
// tiles (its 64-KiB prologue fill and LDS-patch epilogue are a third of such a launch), fine for this kernel (continuous operand stream is moot with one
// tile per block, but the in-register epilogue is not).  VLATOUCH_PT_EXTRA=0 for A/B.
bool vt_gemm_pt_extra_shape(const VtGemmParams& p) {
  static const bool on = [] { const char* e = getenv("VLATOUCH_PT_EXTRA"); return !e || atoi(e) != 0; }();
  if (!on || !pt_enabled() || p.cmap != 0 || !vt_gemm_fast_eligible(p) || !pt_kind_ok(p)) return false;
  const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  return tiles256 >= 160 && tiles256 <= 256 && p.K >= 512;
}

// which launches take the persistent kernel
bool vt_gemm_pt_eligible(const VtGemmParams& p) { return pt_enabled() && pt_kind_ok(p) && (vt_gemm_pp_shape(p) || vt_gemm_pt_extra_shape(p)); }

int vt_gemm_pt_launch(const VtGemmParams& p, hipStream_t s) {
  if (!g_pt_cus) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return VT_ERR_LAUNCH;
    g_pt_cus = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  const bool kv = p.cmap == 3;
  const int tiles_n = ((kv ? p.N / 2 : p.N) + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;     // KV: tiles of ONE half (role)
  const int total = tiles_n * tiles_m;
  // super-row height of the tile order: the 16 blocks of one role on an XCD then cover 4 m-tiles x 4 n-tiles at a time (KV kind: 8 n-tiles per role;
  // measured 1 812 us at 4, 1 831 at 8, 1 864 at 16 on the K|V shape); the other kinds keep 8
  const int gm = g_vt_gm > 0 ? g_vt_gm : (kv ? 4 : 8);
  int grid;
  // bits 8..11 of the last argument: which bit of blockIdx.x selects the role of a KV-kind block (3 .. 7; VLATOUCH_PT_ROLE_BIT for A/B)
  static const int role_bit = [] { const char* e = getenv("VLATOUCH_PT_ROLE_BIT"); const int v = e ? atoi(e) : 3; return v >= 3 && v <= 7 ? v : 3; }();
  static const int kv_split = [] { const char* e = getenv("VLATOUCH_PT_KV_SPLIT"); return e ? atoi(e) : 1; }();
  if (kv) {
    static const int kv_grid = [] { const char* e = getenv("VLATOUCH_PT_KV_GRID"); return e ? atoi(e) : 0; }();   // A/B: leave CUs to a co-running stream
    const int cap = kv_grid > 0 && kv_grid < g_pt_cus ? kv_grid : g_pt_cus;
    if (kv_split) {                                    // two roles: whole groups of 256 blocks (role bit up to 7)
      grid = 2 * total < cap ? 2 * total : cap;
      grid &= ~((2 << role_bit) - 1);                  // whole groups of 2^(role_bit + 1) blocks: 16 for the default bit 3
      if (grid < (2 << role_bit)) return VT_ERR_UNSUPPORTED;
    } else {                                           // every block walks both halves: one block per CU, XCD-banded
      grid = total < cap ? total : cap;
      if (grid >= 8) grid &= ~7;
    }
  } else {
    // one block per tile when the tiles fit the chip (195 tiles must not become 192 blocks + a second round for 3 of them: the kernel falls back to the
    // plain tile order when the grid is not a multiple of 8); otherwise one block per CU, XCD-banded
    grid = total <= g_pt_cus ? total : (g_pt_cus & ~7);
  }
  VtProfScope prof(2, p, s);
#ifdef VLATOUCH_BENCH_BUILD
  { const char* e = getenv("VLATOUCH_PT_ABL"); g_pt_abl = e ? atoi(e) : 0; }
#endif
#define VT_PT_GO(T16, KIND, ACT) hipLaunchKernelGGL((gemm_pt_kernel<T16, KIND, ACT>), dim3(grid), dim3(512), 0, s, p, tiles_n, tiles_m, total, gm, g_pt_abl | (role_bit << 8) | (kv_split ? (1 << 12) : 0))
  if (p.cmap == 3) { if (p.a_dtype == VT_BF16) VT_PT_GO(bf16_t, KIND_KV, VT_ACT_NONE); else VT_PT_GO(half_t, KIND_KV, VT_ACT_NONE); }
  else if (p.c_dtype == VT_F32) { if (p.a_dtype == VT_BF16) VT_PT_GO(bf16_t, KIND_R32, VT_ACT_NONE); else VT_PT_GO(half_t, KIND_R32, VT_ACT_NONE); }
  else if (p.a_dtype == VT_BF16) {
    if (p.act == VT_ACT_NONE) VT_PT_GO(bf16_t, KIND_P16, VT_ACT_NONE);
    else if (p.act == VT_ACT_GELU_ERF) VT_PT_GO(bf16_t, KIND_P16, VT_ACT_GELU_ERF);
    else VT_PT_GO(bf16_t, KIND_P16, VT_ACT_GELU_TANH);
  } else {
    if (p.act == VT_ACT_NONE) VT_PT_GO(half_t, KIND_P16, VT_ACT_NONE);
    else if (p.act == VT_ACT_GELU_ERF) VT_PT_GO(half_t, KIND_P16, VT_ACT_GELU_ERF);
    else VT_PT_GO(half_t, KIND_P16, VT_ACT_GELU_TANH);
  }
#undef VT_PT_GO
  return vt_check_launch();
}
