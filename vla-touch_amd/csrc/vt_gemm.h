// vt_gemm.h — parameter block of the generic MFMA GEMM / implicit-conv1d kernel.
//   C[m, n] = epilogue( sum_k A'[m, k] * W[n, k] ),   W is [N, K] row-major (torch Linear layout).
// A' is either A[m, k] (plain) or the implicit im2col of a channel-last sequence tensor
// A[b, t, c] (conv mode):  m = b*Tout + t,  k = tap*Cin + c,
//   A'[m, k] = A[b, t*stride + off0 + tap*tstep, c]  if that row is inside [0, Tin), else 0.
// Grouped (z) launches: z = group*splitk + slice; each group offsets A/W/C/bias/residual by a fixed
// element stride; each split-K slice writes a raw fp32 partial slab (no epilogue) at C + slice*c_slab.
#pragma once
#include <stdint.h>

struct VtGemmParams {
  const void* A; const void* W; void* C;
  int M, N, K;
  long lda, ldw, ldc;
  // conv mode (taps == 0 -> plain GEMM)
  int taps, cin, tout, tin, stride, off0, tstep;
  // epilogue: v = acc + bias[n]; v = act(v); v *= colscale[n]; v += residual[m, n]; store
  const float* bias; const float* colscale; const void* residual; long ldr;
  int act;
  // grouping / split-K
  int groups, splitk;
  long a_gs, w_gs, c_gs, bias_gs, r_gs;   // per-group element strides
  long c_slab;                            // per-slice element stride of the fp32 partial slabs
  int a_dtype, w_dtype, c_dtype;          // VT_F32 / VT_BF16 (residual has c_dtype)
  // fused per-head (64 columns) RMSNorm after the bias — timm Attention q_norm / k_norm (large-GEMM path only):
  // columns [0, hn_c0_end) use gains hn_w0[64], columns [hn_c0_end, hn_c1_end) use hn_w1[64]; mode 1 mean-square, 2 variance.
  const float* hn_w0; const float* hn_w1;
  int hn_c0_end, hn_c1_end;
  float hn_eps; int hn_mode;
  // output mapping of the cached-condition K / V projections (large-GEMM path, 16-bit C, N % 64 == 0, no residual):
  // row m (all samples of the batch back to back), column n = h*64 + d go to the per-head tile stream of vt_attn_kvt.hip,
  //   tile(h, t = m/64) = C + ((h*cmap_T + t) * 2) * 4096 elements:  [K: 64 rows x 64 d][Vt: 64 d x 64 rows in MFMA k order]
  // cmap 0 = plain row-major C; 1 = K part; 2 = Vt part (written transposed from the epilogue patch); 3 = fused K|V projection
  // (N = 2*D: columns [0, D) -> K part, columns [D, 2D) -> Vt part of head (n - D)/64).  cmap_T = ceil(M/64).
  int cmap, cmap_T;
  // optional second copy of W in MFMA fragment order (vt_pack_w32: [N/32][K/16][64 lanes][8], N % 32 == 0, K % 16 == 0) for the
  // weights-in-registers tile of vt_gemm_pw.hip; null = not available.  W itself stays valid (other tile shapes read it).
  const void* Wp;
  // scratch of the small-M tile of vt_gemm_pws.hip when it splits K over blocks: fp32 partial slabs [S][M][N] (sk_ws_bytes available) and
  // one zero-initialised, self-resetting ticket counter per 96 x 64 output tile (sk_cnt_n of them).  Null = that tile keeps S = 1.
  void* sk_ws; size_t sk_ws_bytes; int* sk_cnt; int sk_cnt_n;
  // RMSNorm hand-off between a residual Linear and the Linear that consumes norm(x) * gain (vt_gemm_pw.hip only):
  //   producer (fp32 C = residual + product): besides C it writes xn_out[m][n] = C[m][n] * xn_gain[n] in the 16-bit operand type (row stride xn_ld)
  //     and the PAIR xn_part[m][j = 2 * (n / 128) + (n % 128) / 64][2] = (sum of squares, sum) of C over those 64 columns — no atomics, fixed order;
  //   consumer (A = that xn_out): every accumulator row is multiplied by rstd[m] before the bias — (x * gain) W^T * rstd = (x * rstd * gain) W^T, the
  //     norm launch between the two Linears disappears.  With Q = sum_{j < rs_n} rs_part[m][j][0], S = sum_j rs_part[m][j][1], K = 1 / rs_inv_k:
  //       rs_mode 0 / 1 (mean-square, timm >= 1.0.9):            rstd = rsqrt(Q / K + rs_eps)
  //       rs_mode 2 (unbiased variance, x not centred, <= 1.0.8): rstd = rsqrt(M2 / (K - 1) + rs_eps); for this form the PRODUCER (which is given rs_mode too) writes
  //         xn_part[m][j] = (second moment of the 64 columns about their own mean, sum) and the consumer merges the groups pairwise (Chan et al.) — never Q - S^2 / K
  void* xn_out; long xn_ld; const float* xn_gain; float* xn_part;
  const float* rs_part; int rs_n; float rs_inv_k, rs_eps; int rs_mode;
  // prefetch hint (vt_gemm_pw.hip): pf_bytes of memory at pf_ptr — the NEXT launch's frozen weights — are touched (one dword per 64 bytes, dealt over the
  // blocks) at the top of this launch's epilogue, so that they stream HBM -> Infinity Cache under its store drain and the kernel boundary instead of at the
  // head of the next launch's k-loop.  Null = none.  Results do not depend on it.
  const void* pf_ptr; size_t pf_bytes;
  // range guard (include/vlatouch.h, vt_rdt_set_range_flag): device word the hand-off producer ORs VT_RANGE_XN_SAT into when it clamps; null = none
  unsigned* range_flag;
};
