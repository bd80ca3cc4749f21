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

// ---- 16-bit outputs, 8 columns per lane: the read-back of a 32 x 64 patch is 4 wave instructions of 8 rows (8 lanes x 8 columns cover a
// row's 64 columns) and every lane stores 16 bytes.  With 4 columns per lane the epilogue of a 256 x 256 tile was 256 dwordx2 store
// instructions per CU, and those are issue-bound (~7 B/clk/CU whatever their width, MI355X_MICROARCH.md "epilogue store tail"): the 128 KB of
// a bf16 tile took ~8 us of a 58 us tile.  Same arithmetic as vt_epi_segment, same order (the head-norm sum runs over 8 lanes of 8 squares).
__device__ __forceinline__ float row8_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror: quad 0 <-> quad 1 of each group of 8
  return v;
}
struct VtEpi8Consts { float4 b[2], cs[2], hw[2]; };
template <typename TC, int CMAP, bool ACT>
__device__ __forceinline__ void vt_epi_segment8(const VtGemmParams& p, const float4 x0, const float4 x1, const VtEpi8Consts& k, const float* hw,
                                                TC* Cg, const TC* Rg, const int m, const int n, const int ncol0, const bool col_ok) {   // CMAP 0 or 1
  static_assert(sizeof(TC) == 2, "16-bit outputs");
  float o[8] = {x0.x + k.b[0].x, x0.y + k.b[0].y, x0.z + k.b[0].z, x0.w + k.b[0].w, x1.x + k.b[1].x, x1.y + k.b[1].y, x1.z + k.b[1].z, x1.w + k.b[1].w};
  if (hw) {
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) q += o[r] * o[r];
    q = row8_sum(q);
    float var;
    if (p.hn_mode == 2) {
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sm += o[r];
      const float mean = row8_sum(sm) * (1.f / 64.f);
      var = (q - 64.f * mean * mean) * (1.f / 63.f);
    } else var = q * (1.f / 64.f);
    const float rstd = rsqrtf(var + p.hn_eps);
    const float g8[8] = {k.hw[0].x, k.hw[0].y, k.hw[0].z, k.hw[0].w, k.hw[1].x, k.hw[1].y, k.hw[1].z, k.hw[1].w};
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] *= rstd * g8[r];
  }
  if constexpr (ACT) {
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = act_apply(o[r], p.act);
  }
  const float c8[8] = {k.cs[0].x, k.cs[0].y, k.cs[0].z, k.cs[0].w, k.cs[1].x, k.cs[1].y, k.cs[1].z, k.cs[1].w};
#pragma unroll
  for (int r = 0; r < 8; ++r) o[r] *= c8[r];
  if (m >= p.M || !col_ok) return;
  TC* dst;
  if constexpr (CMAP == 1) dst = Cg + (((long)(ncol0 >> 6) * p.cmap_T + (m >> 6)) * 2) * 4096 + (m & 63) * 64 + (n - ncol0);
  else {
    dst = Cg + (long)m * p.ldc + n;
    if (Rg) {
      TC rv[8];
      *reinterpret_cast<uint4*>(rv) = *reinterpret_cast<const uint4*>(Rg + (long)m * p.ldr + n);
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] += Elem<TC>::to_f(rv[r]);
    }
  }
  TC ov[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) ov[r] = Elem<TC>::from_f(o[r]);
  *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(ov);
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
      // Vt tiles (bias only).  A lane takes the 4 keys kq .. kq+3 of BOTH 16-row tiles of the patch for one d row: in the tile's k order
      // (vt_kpos) the groups k .. k+3 and k+16 .. k+19 are neighbours, so the lane stores 16 contiguous bytes; 4 lanes cover the patch's 32
      // keys of a d row, 16 d rows per instruction.  (A 16-row last patch, or rows past M, fall back to 8-byte / scalar stores.)
      const long tbase = (long)(hcol0 >> 6) * p.cmap_T;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int dd = it * 16 + (lane >> 2), kq = (lane & 3) * 4;
        const float4 x0 = *reinterpret_cast<const float4*>(ep + dd * EPT_LD + kq);
        const float4 x1 = *reinterpret_cast<const float4*>(ep + dd * EPT_LD + 16 + kq);
        const float bv = bias ? bias[ncol0 + dd] : 0.f;
        const int m = mrow0 + jp * 16 + kq;
        if (m >= p.M) continue;
        TC* dst = Cg + ((tbase + (m >> 6)) * 2 + 1) * 4096 + dd * 64 + vt_kpos(m & 63);
        TC ov[8] = {Elem<TC>::from_f(x0.x + bv), Elem<TC>::from_f(x0.y + bv), Elem<TC>::from_f(x0.z + bv), Elem<TC>::from_f(x0.w + bv),
                    Elem<TC>::from_f(x1.x + bv), Elem<TC>::from_f(x1.y + bv), Elem<TC>::from_f(x1.z + bv), Elem<TC>::from_f(x1.w + bv)};
        if (nrows == 32 && m + 19 < p.M && ((mrow0 + jp * 16) & 31) == 0) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(ov);
        else {
          for (int e = 0; e < 4; ++e) if (m + e < p.M) dst[e] = ov[e];
          if (nrows == 32) {
            TC* dst2 = Cg + ((tbase + ((m + 16) >> 6)) * 2 + 1) * 4096 + dd * 64 + vt_kpos((m + 16) & 63);
            for (int e = 0; e < 4; ++e) if (m + 16 + e < p.M) dst2[e] = ov[4 + e];
          }
        }
      }
    } else {
      bool wide = false;
      if constexpr (sizeof(TC) == 2) wide = (p.N % 8) == 0 && (CMAP == 1 || ((p.ldc % 8) == 0 && (!Rg || (p.ldr % 8) == 0)));
      if (wide) {
        if constexpr (sizeof(TC) == 2) {
          // 8 lanes cover one row (8 columns each), 8 rows per instruction, 16-byte stores (see vt_epi_segment8)
          const int c8 = lane & 7, n8 = ncol0 + c8 * 8;
          const bool ok8 = n8 < p.N;
          VtEpi8Consts k;
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            k.b[hf] = (bias && ok8) ? *reinterpret_cast<const float4*>(bias + n8 + hf * 4) : zero4;
            k.cs[hf] = (p.colscale && ok8) ? *reinterpret_cast<const float4*>(p.colscale + n8 + hf * 4) : one4;
            k.hw[hf] = hw ? *reinterpret_cast<const float4*>(hw + c8 * 8 + hf * 4) : one4;
          }
          if (p.act != VT_ACT_NONE) {
#pragma unroll 2
            for (int it = 0; it < nrows / 8; ++it) {
              const int row = it * 8 + (lane >> 3);
              const float4 x0 = *reinterpret_cast<const float4*>(ep + row * EP_LD + c8 * 8), x1 = *reinterpret_cast<const float4*>(ep + row * EP_LD + c8 * 8 + 4);
              vt_epi_segment8<TC, CMAP, true>(p, x0, x1, k, hw, Cg, Rg, mrow0 + jp * 16 + row, n8, ncol0, ok8);
            }
          } else {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              if (it * 8 >= nrows) break;
              const int row = it * 8 + (lane >> 3);
              const float4 x0 = *reinterpret_cast<const float4*>(ep + row * EP_LD + c8 * 8), x1 = *reinterpret_cast<const float4*>(ep + row * EP_LD + c8 * 8 + 4);
              vt_epi_segment8<TC, CMAP, false>(p, x0, x1, k, hw, Cg, Rg, mrow0 + jp * 16 + row, n8, ncol0, ok8);
            }
          }
        }
      } else if (p.act != VT_ACT_NONE) {
        // read the 32 x 64 patch back row-contiguously: 16 lanes cover one row (64 floats), 4 rows per instruction
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
