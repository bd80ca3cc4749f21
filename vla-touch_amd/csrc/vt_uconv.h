// vt_uconv.h — parameter blocks of the fused conditional-U-Net kernels (vt_uconv.hip), internal to libvlatouch_hip.so.
//
// The interpolant sampler's U-Nets are a dependent chain of small convolutions (M = B*T_l <= 512 rows).  In the launch-per-op driver every
// Conv1d is followed by a GroupNorm+Mish(+FiLM / +residual) kernel and every launch costs >= 4.7 us of the chain.  Here an activation tensor
// between two convolutions is never finished by a kernel of its own: it stays DEFERRED — raw split-K slabs + bias + GroupNorm parameters +
// FiLM rows + residual — and the consuming convolution resolves it in its prologue (sum the slabs, group statistics, Mish, FiLM, residual,
// split into bf16 hi / lo MFMA operands in LDS).  The block with n-tile 0 also writes the resolved tensor to HBM where a later launch needs
// it as a plain tensor (identity residuals, skip connections).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct USrc {
  // nslabs == 0: plain fp32 tensor p[net*gs + row*ld + c], channels >= cvalid read as 0.
  // nslabs  > 0: value = bias[c] + sum_s p[net*gs + s*slab + row*ld + c]
  const float* p; long ld, gs; int nslabs; long slab; int cvalid;
  const float* bias; const float* gamma; const float* beta; long vec_gs;   // per-net vectors of the source's channel count (vec_gs apart)
  int cpg;                         // > 0: GroupNorm over (T rows x cpg channels) of a sample, then Mish
  // FiLM after the Mish: y = (fs[film_off + c] + fc[b][film_off + c]) * y + (fs[film_off + film_C + c] + fc[b][film_off + film_C + c]);
  // fs = step part (this SDE step's row), fc = per-sample condition part.  Null film_s = none.
  const float* film_s; const float* film_c; long film_s_gs, film_c_gs, film_ld, film_off; int film_C;
  // residual added last: 0 none, 1 plain tensor, 2 slabs + bias
  int res_mode; const float* res; long res_ld, res_gs; int res_nslabs; long res_slab; const float* res_bias;
  float* mat; long mat_ld, mat_gs; // optional: the resolved tensor, written by the blocks of n-tile 0 / parity 0
  int C;                           // channels this source contributes to the reduction
};

struct UConvParams {
  USrc src[2]; int c_split;        // reduction channels [0, c_split) come from src[0], the rest from src[1] (torch.cat of the up path)
  // weights: bf16 hi / lo split, MFMA fragment order [net][parity][N/64][4 waves][C/32][taps][hi|lo][64 lanes][8]
  const uint16_t* Wp; long w_gs, w_ps; int nc32;
  const uint16_t* Wr; long wr_gs;                // has_res: the 1x1 residual convolution's weights, same order with one tap
  float* out; long out_gs, out_slab, ldc;        // raw fp32 slabs [S][rows][N] of the convolution
  int has_res; float* rout; long rout_gs, rout_slab;   // the 1x1 residual convolution of the same input: N/64 more n-tiles of the same launch
  int nets, npar, B, Tin, Tq, stride, ntaps, omul;     // GEMM row (b, t < Tq) reads input rows t*stride + off[par][tap], writes row (b*Tq + t)*omul + par
  int off[2 * 6];
  int N, cs, S, nsamp, mtiles, ntiles, nw;
  int lds_rstage, lds_stats, lds_par, lds_hi, lds_lo, pitch;    // byte offsets into dynamic LDS; pitch = bytes per operand row
  float eps;
  long long* tbuf;                 // optional phase time stamps (tools/uconv_phases.py): 8 x s_memrealtime per block, null in the product
};

struct UFinalParams {              // final_conv.0's GroupNorm+Mish, final_conv.1 (1x1, C -> dim) of both nets and the Euler-Maruyama update
  const float* slabs; int nslabs; long slab, gs; const float* bias; const float* gamma; const float* beta; long vec_gs; int cpg;
  const float* out_w; const float* out_b; long ow_gs, ob_gs;    // [nets][dim][C], [nets][dim]
  float* x; const float* z; float* traj; float* vs;             // state [B][T][dim] (in place); noise or null; optional copy of the new state; optional raw net outputs [nets][B*T][dim]
  int B, T, C, dim, nets, do_sde;
  float dt, gi, gdg, eps_t, noise_scale, d, score_eps, gn_eps; int backward;
};

int vt_uconv_launch(const UConvParams& p, int J, size_t lds_bytes, hipStream_t s);
// phase time stamps of the following launches: buf[launch][2048 blocks][8] (null = off)
extern "C" int vt_uconv_set_timing(long long* buf, int max_launches);
int vt_ufinal_launch(const UFinalParams& p, hipStream_t s);
size_t vt_ufinal_lds_bytes(int T, int dim, int C);
// fp32 tap-major weights [nets][N][ntaps*cinp] -> the fragment-ordered hi / lo stream described above
int vt_uconv_pack(const float* Wm, uint16_t* out, int nets, int N, int ntaps, int cinp, int nc32, long out_gs, hipStream_t s);
// sinusoidal embedding of n_steps scalar times [n][dsed] (conditional_unet_1D.py:12-19)
int vt_usin_launch(const float* ts_host, int n, float* out, int dsed, hipStream_t s);
