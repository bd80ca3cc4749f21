// vt_gemm_fast.hip — the large-GEMM path: 16-bit x 16-bit -> fp32 accumulate on v_mfma_f32_16x16x32_{bf16,f16}, for every
// big Linear of RDT (bf16) and DINOv2 (IEEE fp16) (C = epilogue(A[M,K] W[N,K]^T), K % 64 == 0).
//
// Structure: BM x 128 x 64 block tile (BM = 128 or 64), 256 threads = 4 waves (2x2), each wave (BM/2) x 64 output =
// (BM/32) x 4 MFMA tiles.  Operand tiles go HBM/L2 -> LDS by DMA (`global_load_lds_dwordx4`, 16 B per lane, no VGPR
// round trip): a wave instruction fills 1 KiB = 8 rows x 128 B of the tile; the LDS image is lane-linear, so the XOR
// chunk swizzle that makes the 16-row fragment reads (ds_read_b128) bank-conflict-free is applied to the SOURCE
// address of each lane and again on the read (same involution both sides).  Two pipelining variants (picked per launch
// from the grid size, see vt_gemm_fast_launch): ONE LDS stage with 4 co-resident blocks per CU (latency hidden across
// blocks; the default whenever the grid fills the 1024 block slots) or TWO stages at 2 blocks/CU (DMA of k-tile t+1 issued
// before the MFMAs of tile t).  Blocks are dealt to XCDs in contiguous bands of tiles, and inside a band in super-rows of
// GM m-tiles, so neighbouring tiles share operand panels in that XCD's private L2.
// Operands are swapped (D = W_tile A_tile^T) so a lane ends with 4 consecutive n of one row m.
// Epilogue: vt_gemm_epilogue.h (shared with the ping-pong kernels vt_gemm_pp.hip / vt_gemm_ppk.hip, which take the shapes
// their tiles fit: this kernel is the fallback for every other large 16-bit GEMM).
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_gemm_epilogue.h"
#include "vt_prof.h"

int g_vt_gm = 0;         // m-tiles per super-row (VLATOUCH_GEMM_GM; 0 = choose per launch)

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BK = 64;

// NS = LDS stages: 2 = the DMA of k-tile t+1 overlaps the MFMAs of tile t inside the block (2 blocks/CU);
//                 1 = no overlap inside a block, latency is hidden by MINW (3-4) co-resident blocks per CU instead.
// Block tile BM x BN: 2 x (BN/64) waves, each (BM/2) x 64 -> 128-col tiles run 4 waves (256 threads), 256-col tiles 8 waves.
// MINW = co-resident blocks per CU the register budget is sized for.
template <typename T16, typename TC, int BM, int BN, int NS, int MINW, int CMAP>
__global__ __launch_bounds__(2 * BN, MINW * (BN / 128)) void gemm_glds_kernel(const VtGemmParams p, const int tiles_n, const int tiles_per_group, const int total_tiles, const int GM) {
  constexpr int WN = BN / 64, NW = 2 * WN;  // waves along N, waves per block (blockDim.x = 64 * NW = 2 * BN)
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int TM = BM / 32;               // 16-row MFMA tiles per wave along M
  constexpr int QA = BM / 8 / NW;           // A-tile DMA instructions per wave (8 rows x 128 B each)
  constexpr int QB = BN / 8 / NW;           // W-tile DMA instructions per wave
  constexpr int SMEM = NS * STAGE_BYTES > NW * EP_BYTES ? NS * STAGE_BYTES : NW * EP_BYTES;   // operand stages, reused by the epilogue patches
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, l15 = lane & 15;

  int bid = blockIdx.x;
  if ((total_tiles & 7) == 0) bid = (bid & 7) * (total_tiles >> 3) + (bid >> 3);     // XCD b%8 gets a contiguous band
  const int grp = bid / tiles_per_group;
  const int t_in = bid - grp * tiles_per_group;
  // tile order inside a group: super-rows of GM m-tiles; within a super-row n advances slowly and m fastest, so GM
  // consecutive blocks share one W tile and the GM A panels stay L2-resident across all n (W is re-read tiles_m/GM times
  // instead of tiles_m times; PMC FETCH showed 2.5x the algorithmic bytes with the plain n-fastest order).
  const int tiles_m = tiles_per_group / tiles_n;
  const int sr = t_in / (GM * tiles_n);
  const int gm = min(GM, tiles_m - sr * GM);
  const int r_in = t_in - sr * GM * tiles_n;
  const int tn = r_in / gm, tm = sr * GM + (r_in - tn * gm);
  const int m0 = tm * BM, n0 = tn * BN;

  const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (long)grp * p.a_gs;
  const uint16_t* W = reinterpret_cast<const uint16_t*>(p.W) + (long)grp * p.w_gs;

  // DMA sources: instruction q of this wave fills 8 LDS rows; lane -> (row, chunk position); it fetches the chunk whose
  // swizzled position is its own.  Rows beyond M / N are clamped (computed, never stored).
  const uint16_t* a_src[QA];
  const uint16_t* b_src[QB];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int r = (wave * QA + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    a_src[q] = A + (long)min(m0 + r, p.M - 1) * p.lda + c * 8;
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int r = (wave * QB + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[q] = W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < QA; ++q)
      __builtin_amdgcn_global_load_lds((glb_void*)(a_src[q] + (long)kt * BK), (lds_void*)(base + (wave * QA + q) * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < QB; ++q)
      __builtin_amdgcn_global_load_lds((glb_void*)(b_src[q] + (long)kt * BK), (lds_void*)(base + BM * 128 + (wave * QB + q) * 1024), 16, 0, 0);
  };

  float4_t acc[4][TM];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  if constexpr (NS == 2) {
    stage(0, 0);
    __syncthreads();        // drains the DMA (vmcnt) and publishes stage 0
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = NS == 2 ? (kt & 1) : 0;
    if constexpr (NS == 2) {
      if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    } else {
      stage(0, kt);
      __syncthreads();      // tile landed
    }
    const char* As = smem + cur * STAGE_BYTES;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Frag<T16> af[TM], wf[4];
#pragma unroll
      for (int j = 0; j < TM; ++j) lds_frag(af[j], As, wm * (BM / 2) + j * 16 + l15, ks * 4 + g);
#pragma unroll
      for (int i = 0; i < 4; ++i) lds_frag(wf[i], Bs, wn * 64 + i * 16 + l15, ks * 4 + g);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[i], af[j]);
    }
    __syncthreads();        // next stage landed (vmcnt(0) inside) and everyone is done reading `cur`
  }

  // ---------------- epilogue (vt_gemm_epilogue.h); the stages are reused as the waves' private patches
  vt_gemm_epilogue<TC, TM, CMAP>(p, acc, reinterpret_cast<float*>(smem + wave * EP_BYTES), grp, m0 + wm * (BM / 2), n0 + wn * 64, lane);
}

}  // namespace

int g_vt_force_bm = 0;   // tuning hooks: VLATOUCH_GEMM_BM=64|128, VLATOUCH_GEMM_VARIANT=22|14
int g_vt_variant = 0;    // 0 = choose per launch; 22 = two LDS stages at 2 blocks/CU, 14 = one stage at 4 blocks/CU

bool vt_gemm_fast_eligible(const VtGemmParams& p) {
  static const bool init = [] { const char* e = getenv("VLATOUCH_GEMM_BM"); if (e) g_vt_force_bm = atoi(e); e = getenv("VLATOUCH_GEMM_VARIANT"); if (e) g_vt_variant = atoi(e); e = getenv("VLATOUCH_GEMM_GM"); if (e) g_vt_gm = atoi(e); return true; }();
  (void)init;
  if ((p.a_dtype != VT_BF16 && p.a_dtype != VT_F16) || p.w_dtype != p.a_dtype || p.taps != 0 || p.splitk != 1) return false;
  if (p.c_dtype != p.a_dtype && p.c_dtype != VT_F32) return false;
  if (p.K % BK || p.lda % 8 || p.ldw % 8 || p.N % 4 || p.ldc % 4 || (p.residual && p.ldr % 4)) return false;
  if (p.M < 128) return false;
  if (p.cmap && ((p.a_dtype != VT_BF16 && !(p.a_dtype == VT_F16 && p.cmap == 3)) || p.c_dtype == VT_F32 || p.N % 64 || p.residual || p.groups != 1 || (long)p.cmap_T * 64 < p.M)) return false;
  if (p.cmap == 3 && (p.N % 128)) return false;      // fused K|V: two halves of whole heads
  const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.groups;
  return tiles >= 96;
}

bool vt_gemm_can_fuse_headnorm(const VtGemmParams& p) { return vt_gemm_fast_eligible(p) && (p.N % 64) == 0; }

template <typename T16, typename TC, int BM, int CMAP>
static void launch_variant(int variant, dim3 grid, hipStream_t s, const VtGemmParams& p, int tiles_n, int per_group, int total) {
  // super-row height (measured): tall-skinny outputs (few n-tiles, many m-tiles: the condition K/V projections) like 16,
  // everything else 4
  const int gm = g_vt_gm > 0 ? g_vt_gm : ((tiles_n <= 16 && per_group / tiles_n >= 128) ? 16 : 4);
  if (variant == 14) hipLaunchKernelGGL((gemm_glds_kernel<T16, TC, BM, 128, 1, 4, CMAP>), grid, dim3(256), 0, s, p, tiles_n, per_group, total, gm);
  else hipLaunchKernelGGL((gemm_glds_kernel<T16, TC, BM, 128, 2, 2, CMAP>), grid, dim3(256), 0, s, p, tiles_n, per_group, total, gm);
}

int vt_gemm_fast_launch(const VtGemmParams& p, hipStream_t s) {
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.groups;
  // Measured on MI355X (tools/gemm_bench.py).  128-column tiles: one LDS stage with 4 co-resident blocks per CU (1024 block
  // slots) beats in-block double buffering at 2 blocks/CU on every shape of this path (cond-K/V 561 -> 854 TF/s, K=768 DINOv2
  // GEMMs 330 -> 490) except when the grid cannot fill the slots, where the two-stage kernel hides latency inside the block;
  // 128-row tiles when they fill the slots, else 64-row tiles.  GEMMs with several rounds of 256-square tiles go to the
  // ping-pong kernel of vt_gemm_pp.hip (half the L2 -> LDS bytes per flop).
  // A ragged last row block that costs a whole extra round of 256-square tiles (DINOv2-base: 64 images x 257 tokens = 64 x 256 + 64
  // rows; fc1's 65 x 12 = 780 tiles are 3.05 rounds of the 256 CUs): the full row blocks go to the ping-pong kernel, the <= 64 remaining
  // rows to a second small launch (rows are independent: an exact row split).  VLATOUCH_GEMM_ROWSPLIT=0 for A/B.
  if (g_vt_force_bm == 0 && vt_gemm_pw_eligible(p)) return vt_gemm_pw_launch(p, s);     // frozen, fragment-packed weights: W never touches LDS
  if (g_vt_force_bm == 0 && p.cmap == 0 && p.groups == 1 && !p.hn_w0 && !p.hn_w1 && p.M % 256 != 0 && p.M % 256 <= 64 && vt_gemm_pp_eligible(p)) {
    static const bool on = [] { const char* e = getenv("VLATOUCH_GEMM_ROWSPLIT"); return !e || atoi(e) != 0; }();
    const long tm = (p.M + 255) / 256, tn = (p.N + 255) / 256;
    VtGemmParams a = p;
    a.M = (int)((tm - 1) * 256);
    if (on && tm > 1 && (tm * tn + 255) / 256 > ((tm - 1) * tn + 255) / 256 && vt_gemm_pp_eligible(a)) {
      VtGemmParams b = p;
      const size_t ea = p.a_dtype == VT_F32 ? 4 : 2, ec = p.c_dtype == VT_F32 ? 4 : 2;
      b.M = p.M - a.M;
      b.A = (const char*)p.A + (size_t)a.M * p.lda * ea;
      b.C = (char*)p.C + (size_t)a.M * p.ldc * ec;
      if (p.residual) b.residual = (const char*)p.residual + (size_t)a.M * p.ldr * ec;
      const int rc = vt_gemm_pp_launch(a, s);
      return rc != VT_OK ? rc : vt_gemm_launch(b, s);
    }
  }
  if (g_vt_force_bm == 256 || (g_vt_force_bm == 0 && vt_gemm_pp_eligible(p))) return vt_gemm_pp_launch(p, s);
  if ((g_vt_force_bm == 160 || g_vt_force_bm == 0) && vt_gemm_ppk_eligible(p)) return vt_gemm_ppk_launch(p, s);
  int bm = tiles128 < 1024 ? 64 : 128;
  if (g_vt_force_bm == 64 || g_vt_force_bm == 128) bm = g_vt_force_bm;
  const int bn = 128;
  const int tiles_n = (p.N + bn - 1) / bn, tiles_m = (p.M + bm - 1) / bm;
  const int per_group = tiles_n * tiles_m, total = per_group * p.groups;
  const int variant = g_vt_variant ? g_vt_variant : (total < 768 ? 22 : 14);   // 22 = two stages, 2 blocks/CU; 14 = one stage, 4 blocks/CU
  VtProfScope prof(3, p, s);
#define VT_FAST_GO(T16, TC, BMv) launch_variant<T16, TC, BMv, 0>(variant, dim3(total), s, p, tiles_n, per_group, total)
#define VT_FAST_GO3(T16, TC) \
  { if (bm == 128) VT_FAST_GO(T16, TC, 128); else VT_FAST_GO(T16, TC, 64); }
#define VT_FAST_GO_CMAP(BMv) \
  { if (p.a_dtype == VT_F16) launch_variant<half_t, half_t, BMv, 3>(variant, dim3(total), s, p, tiles_n, per_group, total);      /* fp16: the fused K|V form only */ \
    else if (p.cmap == 1) launch_variant<bf16_t, bf16_t, BMv, 1>(variant, dim3(total), s, p, tiles_n, per_group, total); \
    else if (p.cmap == 2) launch_variant<bf16_t, bf16_t, BMv, 2>(variant, dim3(total), s, p, tiles_n, per_group, total); \
    else launch_variant<bf16_t, bf16_t, BMv, 3>(variant, dim3(total), s, p, tiles_n, per_group, total); }
  const bool c16 = p.c_dtype != VT_F32;
  if (p.cmap) {
    if (bm == 128) VT_FAST_GO_CMAP(128) else VT_FAST_GO_CMAP(64)
  } else if (p.a_dtype == VT_BF16) {
    if (c16) VT_FAST_GO3(bf16_t, bf16_t) else VT_FAST_GO3(bf16_t, float)
  } else {
    if (c16) VT_FAST_GO3(half_t, half_t) else VT_FAST_GO3(half_t, float)
  }
#undef VT_FAST_GO
#undef VT_FAST_GO3
#undef VT_FAST_GO_CMAP
  return vt_check_launch();
}
