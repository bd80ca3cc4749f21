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
// Epilogue: + bias, optional per-head RMSNorm (q_norm / k_norm of timm Attention: a wave's 64 columns are exactly
// one head, the row's 64 values live in the 4 lanes sharing lane&15 -> two shuffles), activation, column scale
// (LayerScale); then the wave's sub-tile goes through a private LDS patch and is written (and the residual read) as
// WHOLE 256-B / 128-B row segments — the MFMA register layout alone would scatter 32-B pieces over 16 rows per store.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_kernels.h"
#include "vt_prof.h"

int g_vt_gm = 0;         // m-tiles per super-row (VLATOUCH_GEMM_GM; 0 = choose per launch)

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BN = 128, BK = 64;
constexpr int EP_LD = 68;                      // floats per row of the epilogue patch (64 + 4 pad, keeps 16-B alignment)
constexpr int EPT_LD = 36;                     // transposed patch (cmap 2): 64 d-rows x 32 keys + 4 pad
constexpr int EP_BYTES = 64 * EPT_LD * 4;      // per-wave patch: max(32 x EP_LD, 64 x EPT_LD) floats

// NS = LDS stages: 2 = the DMA of k-tile t+1 overlaps the MFMAs of tile t inside the block (2 blocks/CU);
//                 1 = no overlap inside a block, latency is hidden by MINW (3-4) co-resident blocks per CU instead.
template <typename T16, typename TC, int BM, int NS, int MINW, int CMAP>
__global__ __launch_bounds__(256, MINW) void gemm_glds_kernel(const VtGemmParams p, const int tiles_n, const int tiles_per_group, const int total_tiles, const int GM) {
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int TM = BM / 32;               // 16-row MFMA tiles per wave along M
  constexpr int QA = BM / 32;               // A-tile DMA instructions per wave
  constexpr int SMEM = NS * STAGE_BYTES > 4 * EP_BYTES ? NS * STAGE_BYTES : 4 * EP_BYTES;   // operand stages, reused by the epilogue patches
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  const uint16_t* b_src[4];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    const int r = (wave * QA + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    a_src[q] = A + (long)min(m0 + r, p.M - 1) * p.lda + c * 8;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = (wave * 4 + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    b_src[q] = W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < QA; ++q)
      __builtin_amdgcn_global_load_lds((glb_void*)(a_src[q] + (long)kt * BK), (lds_void*)(base + (wave * QA + q) * 1024), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((glb_void*)(b_src[q] + (long)kt * BK), (lds_void*)(base + BM * 128 + (wave * 4 + q) * 1024), 16, 0, 0);
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

  // ---------------- epilogue.  Register layout: C[m = .. + j*16 + l15][n = ncol0 + i*16 + g*4 + r]
  const float* bias = p.bias ? p.bias + (long)grp * p.bias_gs : nullptr;
  const float* cs = p.colscale;
  const int ncol0 = n0 + wn * 64;                      // this wave's 64 columns = one attention head when hn is active
  const int mrow0 = m0 + wm * (BM / 2);
  const float* hw = nullptr;
  if (p.hn_w0 && ncol0 < p.hn_c0_end) hw = p.hn_w0;
  else if (p.hn_w1 && ncol0 >= p.hn_c0_end && ncol0 < p.hn_c1_end) hw = p.hn_w1;
  float* ep = reinterpret_cast<float*>(smem + wave * EP_BYTES);      // private patch: no block barrier needed below
  TC* Cg = reinterpret_cast<TC*>(p.C) + (long)grp * p.c_gs;
  const TC* Rg = p.residual ? reinterpret_cast<const TC*>(p.residual) + (long)grp * p.r_gs : nullptr;
#pragma unroll
  for (int jp = 0; jp < TM; jp += 2) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = jp + jj;
      float v[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nn = min(ncol0 + i * 16 + g * 4 + r, p.N - 1);
          v[i][r] = acc[i][j][r] + (bias ? bias[nn] : 0.f);
        }
      if (hw) {   // per-head RMSNorm over the 64 columns of this row (wave-uniform branch; all lanes shuffle)
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s += v[i][r]; q += v[i][r] * v[i][r]; }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        float var;
        if (p.hn_mode == 2) { const float mean = s * (1.f / 64.f); var = (q - 64.f * mean * mean) * (1.f / 63.f); }
        else var = q * (1.f / 64.f);
        const float rstd = rsqrtf(var + p.hn_eps);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[i][r] = v[i][r] * rstd * hw[i * 16 + g * 4 + r];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 o;
        float* op = &o.x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = act_apply(v[i][r], p.act);
          if (cs) x *= cs[min(ncol0 + i * 16 + g * 4 + r, p.N - 1)];
          op[r] = x;
        }
        if constexpr (CMAP == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ep[(i * 16 + g * 4 + r) * EPT_LD + jj * 16 + l15] = op[r];
        } else {
          *reinterpret_cast<float4*>(ep + (jj * 16 + l15) * EP_LD + i * 16 + g * 4) = o;
        }
      }
    }
    if constexpr (sizeof(TC) == 2 && CMAP == 2) {
      {
        // Vt tiles: 8 lanes cover the 32 keys of one d row (64 B), 8 d rows per instruction.  A lane's 4 keys are
        // consecutive rows m of the GEMM; they leave the fast path when they cross a batch or 64-key tile boundary.
        const long hbase = (long)(ncol0 >> 6);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int dd = it * 8 + (lane >> 3), kq = (lane & 7) * 4;
          const float4 x = *reinterpret_cast<const float4*>(ep + dd * EPT_LD + kq);
          const float o[4] = {x.x, x.y, x.z, x.w};
          const int m = mrow0 + jp * 16 + kq;
          if (m < p.M) {
            const int bb = m / p.cmap_L, l = m - bb * p.cmap_L;
            if (m + 3 < p.M && l + 3 < p.cmap_L && (l & 63) <= 60 && (l & 1) == 0) {
              TC* dst = Cg + ((((long)bb * p.cmap_H + hbase) * p.cmap_T + (l >> 6)) * 2 + 1) * 4096 + dd * 64;
              TC ov[4] = {Elem<TC>::from_f(o[0]), Elem<TC>::from_f(o[1]), Elem<TC>::from_f(o[2]), Elem<TC>::from_f(o[3])};
              const uint32_t* ow = reinterpret_cast<const uint32_t*>(ov);
              *reinterpret_cast<uint32_t*>(dst + vt_kpos(l & 63)) = ow[0];             // aligned key pairs stay adjacent in k order
              *reinterpret_cast<uint32_t*>(dst + vt_kpos((l + 2) & 63)) = ow[1];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int mm = m + e;
                if (mm < p.M) {
                  const int b2 = mm / p.cmap_L, l2 = mm - b2 * p.cmap_L;
                  Cg[((((long)b2 * p.cmap_H + hbase) * p.cmap_T + (l2 >> 6)) * 2 + 1) * 4096 + dd * 64 + vt_kpos(l2 & 63)] = Elem<TC>::from_f(o[e]);
                }
              }
            }
          }
        }
        continue;
      }
    }
    // read the 32 x 64 patch back row-contiguously: 16 lanes cover one row (64 floats), 4 rows per instruction
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + (lane >> 4), c4 = lane & 15;
      const float4 x = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4 * 4);
      const int m = mrow0 + jp * 16 + row, n = ncol0 + c4 * 4;
      if (m < p.M && n < p.N) {
        float o[4] = {x.x, x.y, x.z, x.w};
        if constexpr (sizeof(TC) == 2 && CMAP == 1) {
          {      // K tiles: the wave's 64 columns are one head's row of the tile
            const int bb = m / p.cmap_L, l = m - bb * p.cmap_L;
            TC ov[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = Elem<TC>::from_f(o[r]);
            *reinterpret_cast<uint2*>(Cg + ((((long)bb * p.cmap_H + (ncol0 >> 6)) * p.cmap_T + (l >> 6)) * 2) * 4096 + (l & 63) * 64 + c4 * 4) =
                *reinterpret_cast<const uint2*>(ov);
            continue;
          }
        }
        if constexpr (sizeof(TC) == 4) {
          if (Rg) { const float4 rv = *reinterpret_cast<const float4*>(Rg + (long)m * p.ldr + n); o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w; }
          *reinterpret_cast<float4*>(Cg + (long)m * p.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          if (Rg) {
            TC rv[4];
            *reinterpret_cast<uint2*>(rv) = *reinterpret_cast<const uint2*>(Rg + (long)m * p.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += Elem<TC>::to_f(rv[r]);
          }
          TC ov[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ov[r] = Elem<TC>::from_f(o[r]);
          *reinterpret_cast<uint2*>(Cg + (long)m * p.ldc + n) = *reinterpret_cast<const uint2*>(ov);
        }
      }
    }
  }
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
  if (p.cmap && (p.a_dtype != VT_BF16 || p.c_dtype == VT_F32 || p.N % 64 || p.residual || p.groups != 1 || p.cmap_L <= 0 || p.cmap_T * 64 < p.cmap_L || p.M % p.cmap_L)) return false;
  const long tiles = (long)((p.M + 127) / 128) * ((p.N + BN - 1) / BN) * p.groups;
  return tiles >= 96;
}

bool vt_gemm_can_fuse_headnorm(const VtGemmParams& p) { return vt_gemm_fast_eligible(p) && (p.N % 64) == 0; }

template <typename T16, typename TC, int BM, int CMAP>
static void launch_variant(int variant, dim3 grid, hipStream_t s, const VtGemmParams& p, int tiles_n, int per_group, int total) {
  // super-row height (measured): tall-skinny outputs (few n-tiles, many m-tiles: the condition K/V projections) like 16,
  // everything else 4
  const int gm = g_vt_gm > 0 ? g_vt_gm : ((tiles_n <= 16 && per_group / tiles_n >= 128) ? 16 : 4);
  if (variant == 14) hipLaunchKernelGGL((gemm_glds_kernel<T16, TC, BM, 1, 4, CMAP>), grid, dim3(256), 0, s, p, tiles_n, per_group, total, gm);
  else hipLaunchKernelGGL((gemm_glds_kernel<T16, TC, BM, 2, 2, CMAP>), grid, dim3(256), 0, s, p, tiles_n, per_group, total, gm);
}

int vt_gemm_fast_launch(const VtGemmParams& p, hipStream_t s) {
  const int tiles_n = (p.N + BN - 1) / BN;
  const long tiles128 = (long)((p.M + 127) / 128) * tiles_n * p.groups;
  // Measured on MI355X (tools/gemm_bench.py): one LDS stage with 4 co-resident blocks per CU (1024 block slots) beats
  // in-block double buffering at 2 blocks/CU on every shape of this path (cond-K/V 561 -> 854 TF/s, K=768 DINOv2 GEMMs
  // 330 -> 490) except when the grid cannot fill the slots, where the two-stage kernel hides latency inside the block.
  // 128-row tiles when they fill the slots, else 64-row tiles.
  const int bm = (g_vt_force_bm == 64 || g_vt_force_bm == 128) ? g_vt_force_bm : (tiles128 < 1024 ? 64 : 128);
  const int tiles_m = (p.M + bm - 1) / bm;
  const int per_group = tiles_n * tiles_m, total = per_group * p.groups;
  const int variant = g_vt_variant ? g_vt_variant : (total < 768 ? 22 : 14);   // 22 = two stages, 2 blocks/CU; 14 = one stage, 4 blocks/CU
  VtProfScope prof(true, p, s);
#define VT_FAST_GO(T16, TC, BMv) launch_variant<T16, TC, BMv, 0>(variant, dim3(total), s, p, tiles_n, per_group, total)
#define VT_FAST_GO_CMAP(BMv) \
  { if (p.cmap == 1) launch_variant<bf16_t, bf16_t, BMv, 1>(variant, dim3(total), s, p, tiles_n, per_group, total); \
    else launch_variant<bf16_t, bf16_t, BMv, 2>(variant, dim3(total), s, p, tiles_n, per_group, total); }
  const bool c16 = p.c_dtype != VT_F32;
  if (p.cmap) {
    if (bm == 128) VT_FAST_GO_CMAP(128) else VT_FAST_GO_CMAP(64)
  } else if (p.a_dtype == VT_BF16) {
    if (bm == 128) { if (c16) VT_FAST_GO(bf16_t, bf16_t, 128); else VT_FAST_GO(bf16_t, float, 128); }
    else           { if (c16) VT_FAST_GO(bf16_t, bf16_t, 64);  else VT_FAST_GO(bf16_t, float, 64); }
  } else {
    if (bm == 128) { if (c16) VT_FAST_GO(half_t, half_t, 128); else VT_FAST_GO(half_t, float, 128); }
    else           { if (c16) VT_FAST_GO(half_t, half_t, 64);  else VT_FAST_GO(half_t, float, 64); }
  }
#undef VT_FAST_GO
#undef VT_FAST_GO_CMAP
  return vt_check_launch();
}
