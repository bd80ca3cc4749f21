// vt_gemm_epilogue.h — shared epilogue of the LDS-DMA GEMM kernels (vt_gemm_fast.hip, vt_gemm_pp.hip).
// A wave owns (TM*16) rows x 64 columns; register layout: acc[i][j][r] = C[mrow0 + j*16 + l15][ncol0 + i*16 + g*4 + r].
// + bias, optional per-head RMSNorm (q_norm / k_norm of timm Attention: the wave's 64 columns are exactly one head, the row's
// 64 values live in the 4 lanes sharing lane&15 -> two shuffles), activation, column scale (LayerScale); then the sub-tile
// goes through the wave's private LDS patch `ep` and is written (and the residual read) as WHOLE row segments — the MFMA
// register layout alone would scatter 32-B pieces over 16 rows per store.  CMAP 1 / 2: the K / Vt tile-stream outputs of the
// cached-condition projections (vt_gemm.h).
#pragma once
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_kernels.h"

// global stores of the row-segment epilogue.  VT_EPI_ST_POLICY (set by a translation unit before including this header; A/B only): 1 = nt, 2 = sc0 sc1
// (write-through): a short kernel that stores its whole output in its last microsecond leaves it dirty in L2 for the kernel boundary to write back
#ifndef VT_EPI_ST_POLICY
#define VT_EPI_ST_POLICY 0
#endif
__device__ __forceinline__ void vt_epi_st128(void* ptr, const float4 v) {
#if VT_EPI_ST_POLICY == 1
  typedef __attribute__((ext_vector_type(4))) float f4v;
  __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(ptr));
#elif VT_EPI_ST_POLICY == 2
  typedef __attribute__((ext_vector_type(4))) float f4v;
  const f4v t = {v.x, v.y, v.z, v.w};
  // s_nop: a > 8-byte store reads its data registers a few cycles after issue; hipcc inserts that wait state behind its own stores but does not look inside inline
  // asm (tools/ubench/band_seam.hip, round 6: without it the next VALU write corrupted 20 % of the stored words)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(ptr), "v"(t) : "memory");
#else
  *reinterpret_cast<float4*>(ptr) = v;
#endif
}
__device__ __forceinline__ void vt_epi_st64(void* ptr, const uint2 v) {
#if VT_EPI_ST_POLICY == 1
  typedef __attribute__((ext_vector_type(2))) unsigned u2v;
  __builtin_nontemporal_store((u2v){v.x, v.y}, reinterpret_cast<u2v*>(ptr));
#elif VT_EPI_ST_POLICY == 2
  typedef __attribute__((ext_vector_type(2))) unsigned u2v;
  const u2v t = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(t) : "memory");
#else
  *reinterpret_cast<uint2*>(ptr) = v;
#endif
}

constexpr int EP_LD = 68;                      // floats per row of the epilogue patch (64 + 4 pad, keeps 16-B alignment)
constexpr int EPT_LD = 36;                     // transposed patch (cmap 2): 64 d-rows x 32 keys + 4 pad
constexpr int EP_BYTES = 64 * EPT_LD * 4;      // per-wave patch: max(32 x EP_LD, 64 x EPT_LD) floats

// one row segment of the read-back: 4 consecutive columns n..n+3 of row m, raw accumulator values in x
// rpre: the residual values of this segment when the caller loaded them ahead of time (fp32 output only; then Rg must be null)
template <typename TC, int CMAP, bool ACT>
__device__ __forceinline__ void vt_epi_segment(const VtGemmParams& p, float4 x, const float4 b4, const float4 cs4, const float* hw, const float4 hw4,
                                               TC* Cg, const TC* Rg, const int m, const int n, const int ncol0, const bool col_ok,
                                               const float4* rpre = nullptr) {   // CMAP here is 0 or 1
  float o[4] = {x.x + b4.x, x.y + b4.y, x.z + b4.z, x.w + b4.w};
  if (hw) {   // per-head RMSNorm over the row's 64 columns = the 16 lanes sharing lane>>4 (wave-uniform branch; all lanes shuffle)
    float q = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
    q = row16_sum(q);
    float var;
    if (p.hn_mode == 2) {      // the row sum is only needed by the variance form (wave-uniform branch)
      const float mean = row16_sum(o[0] + o[1] + o[2] + o[3]) * (1.f / 64.f);
      var = (q - 64.f * mean * mean) * (1.f / 63.f);
    } else var = q * (1.f / 64.f);
    const float rstd = rsqrtf(var + p.hn_eps);
    o[0] *= rstd * hw4.x; o[1] *= rstd * hw4.y; o[2] *= rstd * hw4.z; o[3] *= rstd * hw4.w;
  }
  if constexpr (ACT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = act_apply(o[r], p.act);
  }
  o[0] *= cs4.x; o[1] *= cs4.y; o[2] *= cs4.z; o[3] *= cs4.w;
  if (m >= p.M || !col_ok) return;
  if constexpr (sizeof(TC) == 2 && CMAP == 1) {      // K tiles: the wave's 64 columns are one head's row of the tile
    TC ov[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = Elem<TC>::from_f(o[r]);
    *reinterpret_cast<uint2*>(Cg + (((long)(ncol0 >> 6) * p.cmap_T + (m >> 6)) * 2) * 4096 + (m & 63) * 64 + (n - ncol0)) =
        *reinterpret_cast<const uint2*>(ov);
  } else if constexpr (sizeof(TC) == 4) {
    if (rpre) { o[0] += rpre->x; o[1] += rpre->y; o[2] += rpre->z; o[3] += rpre->w; }
    else if (Rg) { const float4 rv = *reinterpret_cast<const float4*>(Rg + (long)m * p.ldr + n); o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w; }
    vt_epi_st128(Cg + (long)m * p.ldc + n, make_float4(o[0], o[1], o[2], o[3]));
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
    vt_epi_st64(Cg + (long)m * p.ldc + n, *reinterpret_cast<const uint2*>(ov));
  }
}

// hcol0: the column the tile-stream head index is taken from (== ncol0 except for the V half of a fused K|V projection, CMAP 3)
template <typename TC, int TM, int CMAP>
__device__ __forceinline__ void vt_gemm_epilogue_impl(const VtGemmParams& p, float4_t (&acc)[4][TM], float* ep, const int grp, const int mrow0,
                                                      const int ncol0, const int hcol0, const int lane) {
  const int g = lane >> 4, l15 = lane & 15;
  const float* bias = p.bias ? p.bias + (long)grp * p.bias_gs : nullptr;
  const float* hw = nullptr;
  if (p.hn_w0 && ncol0 < p.hn_c0_end) hw = p.hn_w0;
  else if (p.hn_w1 && ncol0 >= p.hn_c0_end && ncol0 < p.hn_c1_end) hw = p.hn_w1;
  TC* Cg = reinterpret_cast<TC*>(p.C) + (long)grp * p.c_gs;
  const TC* Rg = p.residual ? reinterpret_cast<const TC*>(p.residual) + (long)grp * p.r_gs : nullptr;
  // everything after the accumulators happens on the row-contiguous side of the patch, where a lane owns 4 fixed columns:
  // its bias / column scale / norm gain are loaded once, and the activation code exists once per patch instead of once per
  // accumulator register (inlined per register it made the kernels 10x larger and cost more than the k-loop at K = 2048)
  const int c4 = lane & 15, n = ncol0 + c4 * 4;
  const bool col_ok = n < p.N;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b4 = (bias && col_ok && CMAP != 2) ? *reinterpret_cast<const float4*>(bias + n) : zero4;
  const float4 cs4 = (p.colscale && col_ok) ? *reinterpret_cast<const float4*>(p.colscale + n) : one4;
  const float4 hw4 = hw ? *reinterpret_cast<const float4*>(hw + c4 * 4) : one4;
#pragma unroll
  for (int jp = 0; jp < TM; jp += 2) {
    constexpr int NROWS_FULL = 32;
    const int nrows = (TM - jp >= 2) ? NROWS_FULL : 16;        // odd TM: the last patch holds one 16-row tile
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (jp + jj >= TM) continue;
        const float4_t a = acc[i][jp + jj < TM ? jp + jj : TM - 1];
        if constexpr (CMAP == 2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ep[(i * 16 + g * 4 + r) * EPT_LD + jj * 16 + l15] = a[r];
        } else {
          *reinterpret_cast<float4*>(ep + (jj * 16 + l15) * EP_LD + i * 16 + g * 4) = make_float4(a[0], a[1], a[2], a[3]);
        }
      }
    if constexpr (sizeof(TC) == 2 && CMAP == 2) {
      // Vt tiles (bias only): 8 lanes cover the patch's 32 rows (= 32 keys, half a tile) of one d row as one 64-byte segment,
      // 8 d rows per instruction; a lane's 4 keys are an aligned group, contiguous in the tile's k order (vt_kpos).
      const long tbase = (long)(hcol0 >> 6) * p.cmap_T;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int dd = it * 8 + (lane >> 3), kq = (lane & 7) * 4;
        const float4 x = *reinterpret_cast<const float4*>(ep + dd * EPT_LD + kq);
        const float bv = bias ? bias[ncol0 + dd] : 0.f;
        const int m = mrow0 + jp * 16 + kq;
        if (m < p.M && kq < nrows) {
          TC* dst = Cg + ((tbase + (m >> 6)) * 2 + 1) * 4096 + dd * 64 + vt_kpos(m & 63);
          TC ov[4] = {Elem<TC>::from_f(x.x + bv), Elem<TC>::from_f(x.y + bv), Elem<TC>::from_f(x.z + bv), Elem<TC>::from_f(x.w + bv)};
          if (m + 3 < p.M) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(ov);
          else
            for (int e = 0; e < 4; ++e) if (m + e < p.M) dst[e] = ov[e];
        }
      }
    } else {
      // read the 32 x 64 patch back row-contiguously: 16 lanes cover one row (64 floats), 4 rows per instruction
      if (p.act != VT_ACT_NONE) {
#pragma unroll 2
        for (int it = 0; it < nrows / 4; ++it) {
          const int row = it * 4 + (lane >> 4);
          const float4 x = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4 * 4);
          vt_epi_segment<TC, CMAP, true>(p, x, b4, cs4, hw, hw4, Cg, Rg, mrow0 + jp * 16 + row, n, ncol0, col_ok);
        }
      } else {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          if (it * 4 >= nrows) break;
          const int row = it * 4 + (lane >> 4);
          const float4 x = *reinterpret_cast<const float4*>(ep + row * EP_LD + c4 * 4);
          vt_epi_segment<TC, CMAP, false>(p, x, b4, cs4, hw, hw4, Cg, Rg, mrow0 + jp * 16 + row, n, ncol0, col_ok);
        }
      }
    }
  }
}


// CMAP 3: fused K|V projection of a cached condition (N = 2*D: columns [0, D) are K — bias, k_norm, K tiles —, columns [D, 2D) are
// V — bias, Vt tiles); a wave's 64 columns are one head of one half, so the choice is wave-uniform.
template <typename TC, int TM, int CMAP>
__device__ __forceinline__ void vt_gemm_epilogue(const VtGemmParams& p, float4_t (&acc)[4][TM], float* ep, const int grp, const int mrow0,
                                                 const int ncol0, const int lane) {
  if constexpr (CMAP == 3) {
    const int half = p.N >> 1;
    if (ncol0 < half) vt_gemm_epilogue_impl<TC, TM, 1>(p, acc, ep, grp, mrow0, ncol0, ncol0, lane);
    else vt_gemm_epilogue_impl<TC, TM, 2>(p, acc, ep, grp, mrow0, ncol0, ncol0 - half, lane);
  } else {
    vt_gemm_epilogue_impl<TC, TM, CMAP>(p, acc, ep, grp, mrow0, ncol0, ncol0, lane);
  }
}
