// vt_train.hip — primitives of the controller TRAINING step: what the forward kernels do not already provide to
// differentiate the interpolant controller's U-Nets and MLPs and to update them (replaces torch autograd + optim.AdamW + torch_ema in
// residual_controller/bridge/bridge_model.py:183-246 (losses), bridge_train.py:49-58, 312-334 (AdamW step, EMA update)).
// Everything is fp32, channel-last [B][T][C], and every matrix product of the backward pass goes through vt_gemm (fp32 MFMA):
//   weight gradient  dW[co][tap*Cin + ci] = dY^T [Cout][M] x im2col(X)^T [taps*Cin][M]^T   -> vt_transpose + vt_im2col_t + vt_gemm
//   data gradient    dX = conv(dY (zero-stuffed for strided convs), W flipped / transposed) -> vt_wflip (+ vt_zero_stuff) + conv-mode vt_gemm
// so the kernels here are the data-movement and element-wise pieces: byte / word traffic, HBM- or launch-bound, no MFMA.
#include <math.h>
#include "vt_common.h"
#include "vt_host.h"
#include "../../include/vlatouch.h"

namespace {

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }

// out[(tap*Cin + ci)][b*Tout + t] = x[b][t*stride + off0 + tap][ci]  (0 outside [0, Tin)): the transposed im2col of a conv layer
__global__ void im2col_t_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int Tin, int Tout, int Cin, int taps, int stride, int off0) {
  __shared__ float tile[32][33];
  // block: 32 (k = tap*Cin + ci) x 32 (m = b*Tout + t)
  const int M = B * Tout, K = taps * Cin;
  const int k0 = blockIdx.y * 32, m0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8) {                             // read: m = m0 + r (row of x), k = k0 + tx (contiguous ci)
    const int m = m0 + r, k = k0 + tx;
    float v = 0.f;
    if (m < M && k < K) {
      const int b = m / Tout, t = m - b * Tout, tap = k / Cin, ci = k - tap * Cin;
      const int ti = t * stride + off0 + tap;
      if (ti >= 0 && ti < Tin) v = x[((long)b * Tin + ti) * Cin + ci];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {                             // write: k = k0 + r, m = m0 + tx (contiguous)
    const int k = k0 + r, m = m0 + tx;
    if (k < K && m < M) out[(long)k * M + m] = tile[tx][r];
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int M, int N) {   // [M][N] -> [N][M]
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) { const int m = m0 + r, n = n0 + tx; tile[r][tx] = (m < M && n < N) ? in[(long)m * N + n] : 0.f; }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) { const int n = n0 + r, m = m0 + tx; if (n < N && m < M) out[(long)n * M + m] = tile[tx][r]; }
}

// out[b][2u][c] = x[b][u][c], out[b][2u+1][c] = 0
__global__ void zero_stuff_kernel(const float* __restrict__ x, float* __restrict__ out, long n_out, int T, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_out) return;
  const int c = (int)(i % C);
  const long r = i / C;
  const int t2 = (int)(r % (2 * T));
  const long b = r / (2 * T);
  out[i] = (t2 & 1) ? 0.f : x[((long)b * T + (t2 >> 1)) * C + c];
}

// W [Cout][taps][Cin] -> WT [Cin][taps][Cout] with the taps reversed: the weights of the data-gradient convolution
__global__ void wflip_kernel(const float* __restrict__ W, float* __restrict__ WT, int Cout, int taps, int Cin) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)Cout * taps * Cin) return;
  const int co = (int)(i % Cout);
  const long r = i / Cout;
  const int tp = (int)(r % taps), ci = (int)(r / taps);
  WT[i] = W[((long)co * taps + (taps - 1 - tp)) * Cin + ci];
}

// column sums of [M][N] (row pitch ld) -> out[N] (bias gradients; the [B][C] partials of the GroupNorm gains).  A block owns 32 columns
// (128-B row segments) with 32 row lanes; each lane adds its rows in order, the lanes are combined by a fixed LDS tree: deterministic.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, long ld, float* __restrict__ out, int M, int N, int accumulate) {
  __shared__ float part[32][33];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + c;
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int m = r;
    for (; m + 32 < M; m += 64) { s0 += x[(long)m * ld + n]; s1 += x[(long)(m + 32) * ld + n]; }
    if (m < M) s0 += x[(long)m * ld + n];
  }
  part[r][c] = s0 + s1;
  __syncthreads();
  for (int h = 16; h > 0; h >>= 1) {
    if (r < h) part[r][c] += part[r + h][c];
    __syncthreads();
  }
  if (r == 0 && n < N) out[n] = accumulate ? out[n] + part[0][c] : part[0][c];
}

// out[i] = sum_s slabs[s][i] (+ bias[i % N]), slabs summed in order: the reduction of a split-K GEMM's fp32 partial products
__global__ void slab_sum_kernel(const float* __restrict__ slabs, int S, long n, const float* __restrict__ bias, int N, float* __restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = *reinterpret_cast<const float4*>(slabs + i);
  for (int s = 1; s < S; ++s) {
    const float4 b = *reinterpret_cast<const float4*>(slabs + (long)s * n + i);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (bias) { const int c = (int)(i % N); a.x += bias[c]; a.y += bias[c + 1]; a.z += bias[c + 2]; a.w += bias[c + 3]; }
  *reinterpret_cast<float4*>(out + i) = a;
}

__global__ void add_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] += b[i];
}
// strided copy of a column block: dst[m][0..cols) = src[m][off .. off+cols)  (splitting the gradient of a channel concat, gathering cond)
__global__ void copy_cols_kernel(const float* __restrict__ src, long lds_, int off, float* __restrict__ dst, long ldd, int doff, int rows, int cols, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int c = (int)(i % cols);
  const long m = i / cols;
  const float v = src[m * lds_ + off + c];
  float* d = dst + m * ldd + doff + c;
  *d = accumulate ? *d + v : v;
}

__device__ __forceinline__ float mish_f(float x) { return act_apply(x, VT_ACT_MISH); }
// d/dx [x tanh(softplus(x))] = tanh(sp) + x (1 - tanh(sp)^2) sigmoid(x)
__device__ __forceinline__ float mish_grad(float x) {
  if (x > 20.0f) return 1.0f;
  const float e = expf(x), w = e * (e + 2.0f);           // libm exp and true divisions: gradients are not the place for the ~1 ulp shortcuts
  const float th = w / (w + 2.0f);
  const float sg = e / (1.0f + e);
  return th + x * (1.0f - th * th) * sg;
}
__global__ void mish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = mish_f(x[i]);
}
__global__ void mish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = dy[i] * mish_grad(x[i]);
}

// Backward of  out = film_scale * mish(GroupNorm(c)) + film_bias (+ residual)  for one (sample, group) per block:
//   in : c [B*T][C] (the conv output the forward normalised), gamma, beta, film [B][2C] or null, dout [B*T][C]
//   out: dc [B*T][C]; dgamma_part / dbeta_part [B][C] (summed over b by vt_colsum); dfilm [B][2C] (d scale | d bias) when film
__global__ __launch_bounds__(256) void gn_mish_bwd_kernel(const float* __restrict__ c, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ film, const float* __restrict__ dout, float* __restrict__ dc,
                                                          float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, float* __restrict__ dfilm,
                                                          int B, int T, int C, int ngroups, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];       // xhat [n] | dxhat [n]
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x / ngroups, grp = blockIdx.x - b * ngroups;
  const int cpg = C / ngroups, n = cpg * T, c0 = grp * cpg;
  float* xh = sm;
  float* dxh = sm + n;
  auto block_sum = [&](float t) {
    t = wave_sum(t);
    __syncthreads();
    if (lane == 0) red[wv] = t;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  float s = 0.f;
  for (int e = tid; e < n; e += 256) { const int t = e / cpg, cc = e - t * cpg; const float v = c[((long)b * T + t) * C + c0 + cc]; xh[e] = v; s += v; }
  const float mean = block_sum(s) / (float)n;
  float q = 0.f;
  for (int e = tid; e < n; e += 256) { const float d = xh[e] - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum(q) / (float)n + eps);
  const float* fl = film ? film + (long)b * 2 * C : nullptr;
  float s1 = 0.f, s2 = 0.f;
  for (int e = tid; e < n; e += 256) {
    const int t = e / cpg, cc = e - t * cpg, col = c0 + cc;
    const float xhat = (xh[e] - mean) * rstd;
    const float u = xhat * gamma[col] + beta[col];
    float dh = dout[((long)b * T + t) * C + col];
    if (fl) dh *= fl[col];
    const float du = dh * mish_grad(u);
    const float dx = du * gamma[col];
    xh[e] = xhat; dxh[e] = dx;
    s1 += dx; s2 += dx * xhat;
  }
  const float m1 = block_sum(s1) / (float)n, m2 = block_sum(s2) / (float)n;
  for (int e = tid; e < n; e += 256) {
    const int t = e / cpg, cc = e - t * cpg;
    dc[((long)b * T + t) * C + c0 + cc] = rstd * (dxh[e] - m1 - xh[e] * m2);
  }
  // per-channel sums over t (this sample): d gamma, d beta, d film scale, d film bias — channel cc by thread cc (cpg <= 256)
  for (int cc = tid; cc < cpg; cc += 256) {
    const int col = c0 + cc;
    float dg = 0.f, db = 0.f, dsc = 0.f, dbi = 0.f;
    for (int t = 0; t < T; ++t) {
      const int e = t * cpg + cc;
      const float xhat = xh[e];
      const float u = xhat * gamma[col] + beta[col];
      const float dof = dout[((long)b * T + t) * C + col];
      const float dh = fl ? dof * fl[col] : dof;
      const float du = dh * mish_grad(u);
      dg += du * xhat; db += du;
      if (fl) { dsc += dof * mish_f(u); dbi += dof; }
    }
    dgamma_part[(long)b * C + col] = dg;
    dbeta_part[(long)b * C + col] = db;
    if (fl) { dfilm[(long)b * 2 * C + col] = dsc; dfilm[(long)b * 2 * C + C + col] = dbi; }
  }
}

__device__ __forceinline__ float gelu_grad(float x) {      // d/dx [0.5 x (1 + erf(x / sqrt 2))]
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
__global__ void gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = dy ? dy[i] * gelu_grad(x[i]) : act_apply(x[i], VT_ACT_GELU_ERF);
}

// q_sample + loss targets of the LINEAR interpolant (bridge_model.py:103-107, 148-150, 183-217, 248-258), per element:
//   tc = clip(t_b, t_min, 1 - t_min);  xt = (1 - tc) x0 + tc x1 + gamma(tc) z
//   target_v = x1 - x0;  target_s = -z;  target_b = (x1 - x0) + gamma'(tc) z;   tclip[b] = tc
// gamma_type: 0 = 1.4142 t (1 - t), 1 = 1.4142 sqrt(t (1 - t)), 2 = 1.4142 (1 - t)^2 sqrt(t)   (the constants the reference writes)
// interp (round 6; bridge_model.py:103-147 `interpolant`, :149-181 `interpolant_dev`): xt = w0 x0 + w1 x1 + gamma z, target_v = d(w0 x0 + w1 x1) / dt
//   0 linear                 w0 = 1 - t            w1 = t                 dv = x1 - x0
//   1 power3                 w0 = (1 - t)^3        w1 = 1 - w0            dv = 3 (1 - t)^2 (x1 - x0)
//   2 power4                 w0 = (1 - t)^4        w1 = 1 - w0            dv = 4 (1 - t)^3 (x1 - x0)
//   3 reverse_power3         w0 = 1 - t^3          w1 = t^3               dv = 3 t^2 (x1 - x0)
//   4 reverse_power4         w0 = 1 - t^4          w1 = t^4               dv = 4 t^3 (x1 - x0)
//   5 gaussian_encode_decode w0 = cos^2(pi t) [t <= .5], w1 = cos^2(pi t) [t > .5]     dv = -2 pi cos(pi t) sin(pi t) ([t <= .5] x0 + [t > .5] x1)
//   6 reverse_linear         w0 = (1 - 2t) [t <= .5],  w1 = 1 - w0        dv = 2 [t <= .5] (x1 - x0)
__global__ void si_qsample_kernel(const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ z, const float* __restrict__ t,
                                  float* __restrict__ xt, float* __restrict__ tv, float* __restrict__ ts, float* __restrict__ tb, float* __restrict__ tclip,
                                  int B, long per, int gamma_type, float t_min, int interp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * per) return;
  const int b = (int)(i / per);
  const float tc = fminf(fmaxf(t[b], t_min), 1.0f - t_min);
  float g, gd;
  if (gamma_type == 1) { g = 1.4142f * sqrtf(tc * (1.0f - tc)); gd = (1.0f - 2.0f * tc) / sqrtf(2.0f * (tc - tc * tc) + 1e-4f); }
  else if (gamma_type == 2) { g = 1.4142f * (1.0f - tc) * (1.0f - tc) * sqrtf(tc);
                              gd = 1.4142f * (2.0f * (tc - 1.0f) * sqrtf(tc) + (1.0f - tc) * (1.0f - tc) / (2.0f * sqrtf(tc + 1e-4f))); }
  else { g = 1.4142f * tc * (1.0f - tc); gd = 1.4142f * (1.0f - 2.0f * tc); }
  const float a = x0[i], c = x1[i], zz = z[i];
  float w0 = 1.0f - tc, w1 = tc, dv = c - a;
  if (interp != 0) {
    const float u = 1.0f - tc, lo = tc <= 0.5f ? 1.0f : 0.0f;
    switch (interp) {
      case 1: w0 = u * u * u; w1 = 1.0f - w0; dv = 3.0f * (u * u) * (c - a); break;
      case 2: w0 = powf(u, 4.0f); w1 = 1.0f - w0; dv = 4.0f * (u * u * u) * (c - a); break;
      case 3: w1 = tc * tc * tc; w0 = 1.0f - w1; dv = 3.0f * (tc * tc) * (c - a); break;
      case 4: w1 = powf(tc, 4.0f); w0 = 1.0f - w1; dv = 4.0f * (tc * tc * tc) * (c - a); break;
      case 5: { const float cs = cosf(tc * 3.14159265358979323846f), sn = sinf(3.14159265358979323846f * tc), c2 = cs * cs;
                w0 = c2 * lo; w1 = c2 * (1.0f - lo);
                const float k = -2.0f * 3.14159265358979323846f * cs * sn;
                dv = k * lo * a + k * (1.0f - lo) * c; break; }
      default: w0 = (1.0f - 2.0f * tc) * lo; w1 = 1.0f - w0; dv = -2.0f * lo * a + 2.0f * lo * c; break;
    }
  }
  xt[i] = w0 * a + w1 * c + g * zz;
  tv[i] = dv;
  ts[i] = -zz;
  tb[i] = dv + gd * zz;
  if (i % per == 0) tclip[b] = tc;
}

// interpolant losses (bridge_model.py:183-217): loss = mean_b(0.5 |o_b|^2 - <tgt_b, o_b>), d loss / d o = (o - tgt) / B; one block
__global__ __launch_bounds__(256) void si_loss_kernel(const float* __restrict__ o, const float* __restrict__ tgt, float* __restrict__ dout, float* __restrict__ loss,
                                                      int B, long per) {
  __shared__ float red[4];
  float s = 0.f;
  const long n = (long)B * per;
  const float invB = 1.0f / (float)B;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float ov = o[i], tv = tgt[i];
    s += 0.5f * ov * ov - tv * ov;
    dout[i] = (ov - tv) * invB;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * invB;
}

// One element of torch.optim.AdamW / torch_ema.  Contraction is switched off so that the three kernels below (scalar arguments, scalars from
// device memory, multi-tensor table) round identically: a replayed graph and the eager step then agree bit for bit.
__device__ __forceinline__ float adamw_elem(float p, float gv, float& m, float& v, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
  const float pv = p * (1.0f - lr * wd);
  const float mv = b1 * m + (1.0f - b1) * gv;
  const float vv = b2 * v + (1.0f - b2) * gv * gv;
  m = mv; v = vv;
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  return pv - (lr / bc1) * (mv / denom);
}
__device__ __forceinline__ float ema_elem(float sh, float p, float one_minus_decay) {
#pragma clang fp contract(off)
  return sh - one_minus_decay * (sh - p);
}
// torch.optim.AdamW (decoupled weight decay, bias-corrected moments), one fused pass
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float mv = m[i], vv = v[i];
  p[i] = adamw_elem(p[i], g[i], mv, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
  m[i] = mv; v[i] = vv;
}
// the same update with the step-dependent scalars read from device memory (hyper = [lr, 1 - beta1^t, sqrt(1 - beta2^t), 1 - ema_decay_t]),
// so that a captured hipGraph of the step can be replayed while the step count advances
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                 const float* __restrict__ hyper, float b1, float b2, float eps, float wd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float lr = hyper[0], bc1 = hyper[1], bc2_sqrt = hyper[2];
  float mv = m[i], vv = v[i];
  p[i] = adamw_elem(p[i], g[i], mv, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
  m[i] = mv; v[i] = vv;
}
__global__ void ema_dev_kernel(float* __restrict__ shadow, const float* __restrict__ p, long n, const float* __restrict__ hyper) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) shadow[i] = ema_elem(shadow[i], p[i], hyper[3]);
}
// AdamW (+ EMA where the tensor has a shadow) over a TABLE of tensors in one launch: a training step updates ~380 tensors, most of them
// a few KB, and one launch each is launch-gap-bound.  tab[k] = {p, g, m, v, shadow | null, n, first_chunk}; a block takes one 4096-
// element chunk and finds its tensor by binary search over first_chunk.  Arithmetic identical to adamw_dev_kernel / ema_dev_kernel.
struct MtEntry { float* p; const float* g; float* m; float* v; float* shadow; long n; long first_chunk; };
__global__ __launch_bounds__(256) void adamw_ema_mt_kernel(const MtEntry* __restrict__ tab, int ntensors, const float* __restrict__ hyper,
                                                           float b1, float b2, float eps, float wd) {
  int lo = 0, hi = ntensors - 1;
  const long chunk = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1; }
  const MtEntry e = tab[lo];
  const long base = (chunk - e.first_chunk) * 4096;
  const float lr = hyper[0], bc1 = hyper[1], bc2_sqrt = hyper[2], omd = hyper[3];
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const long i = base + it * 256 + threadIdx.x;
    if (i >= e.n) break;
    float mv = e.m[i], vv = e.v[i];
    const float pv = adamw_elem(e.p[i], e.g[i], mv, vv, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
    e.m[i] = mv; e.v[i] = vv;
    e.p[i] = pv;
    if (e.shadow) e.shadow[i] = ema_elem(e.shadow[i], pv, omd);
  }
}

// torch_ema: shadow -= (1 - decay) * (shadow - p)
__global__ void ema_kernel(float* __restrict__ shadow, const float* __restrict__ p, long n, float one_minus_decay) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) shadow[i] = ema_elem(shadow[i], p[i], one_minus_decay);
}
// SinusoidalPosEmb (conditional_unet_1D.py:7-19): emb[b] = [sin(t_b f_j) | cos(t_b f_j)], f_j = exp(-j log(10000) / (half - 1))
__global__ void posemb_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim, half = dim / 2;
  const int jj = j < half ? j : j - half;
  const float f = expf(-(float)jj * (logf(10000.0f) / (float)(half - 1)));
  const float a = t[b] * f;
  out[i] = j < half ? sinf(a) : cosf(a);
}

}  // namespace

#define LAUNCH_OK() (vt_check_launch())

int vt_im2col_t(const float* x, float* out, int B, int Tin, int Tout, int Cin, int taps, int stride, int off0, vt_stream_t s) {
  if (!x || !out || B < 1 || Tin < 1 || Tout < 1 || Cin < 1 || taps < 1 || stride < 1) return vt_fail(VT_ERR_ARG, "vt_im2col_t: bad argument");
  hipLaunchKernelGGL(im2col_t_kernel, dim3((B * Tout + 31) / 32, (taps * Cin + 31) / 32), dim3(256), 0, (hipStream_t)s, x, out, B, Tin, Tout, Cin, taps, stride, off0);
  return LAUNCH_OK();
}
int vt_transpose(const float* in, float* out, int M, int N, vt_stream_t s) {
  if (!in || !out || M < 1 || N < 1) return vt_fail(VT_ERR_ARG, "vt_transpose: bad argument");
  hipLaunchKernelGGL(transpose_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, (hipStream_t)s, in, out, M, N);
  return LAUNCH_OK();
}
int vt_zero_stuff(const float* x, float* out, int B, int T, int C, vt_stream_t s) {
  if (!x || !out || B < 1 || T < 1 || C < 1) return vt_fail(VT_ERR_ARG, "vt_zero_stuff: bad argument");
  const long n = (long)B * 2 * T * C;
  hipLaunchKernelGGL(zero_stuff_kernel, g1(n), dim3(256), 0, (hipStream_t)s, x, out, n, T, C);
  return LAUNCH_OK();
}
int vt_wflip(const float* W, float* WT, int Cout, int taps, int Cin, vt_stream_t s) {
  if (!W || !WT || Cout < 1 || taps < 1 || Cin < 1) return vt_fail(VT_ERR_ARG, "vt_wflip: bad argument");
  hipLaunchKernelGGL(wflip_kernel, g1((long)Cout * taps * Cin), dim3(256), 0, (hipStream_t)s, W, WT, Cout, taps, Cin);
  return LAUNCH_OK();
}
int vt_colsum(const float* x, long ld, float* out, int M, int N, int accumulate, vt_stream_t s) {
  if (!x || !out || M < 1 || N < 1) return vt_fail(VT_ERR_ARG, "vt_colsum: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 31) / 32), dim3(1024), 0, (hipStream_t)s, x, ld, out, M, N, accumulate);
  return LAUNCH_OK();
}
int vt_slab_sum(const float* slabs, int S, long n, const float* bias, int N, float* out, vt_stream_t s) {
  if (!slabs || !out || S < 1 || n < 4 || n % 4 || N < 4 || N % 4 || n % N) return vt_fail(VT_ERR_ARG, "vt_slab_sum: bad argument (n, N multiples of 4)");
  hipLaunchKernelGGL(slab_sum_kernel, g1(n / 4), dim3(256), 0, (hipStream_t)s, slabs, S, n, bias, N, out);
  return LAUNCH_OK();
}
int vt_add_(float* a, const float* b, long n, vt_stream_t s) {
  if (!a || !b || n < 1) return vt_fail(VT_ERR_ARG, "vt_add_: bad argument");
  hipLaunchKernelGGL(add_kernel, g1(n), dim3(256), 0, (hipStream_t)s, a, b, n);
  return LAUNCH_OK();
}
int vt_copy_cols(const float* src, long lds_, int off, float* dst, long ldd, int doff, int rows, int cols, int accumulate, vt_stream_t s) {
  if (!src || !dst || rows < 1 || cols < 1) return vt_fail(VT_ERR_ARG, "vt_copy_cols: bad argument");
  hipLaunchKernelGGL(copy_cols_kernel, g1((long)rows * cols), dim3(256), 0, (hipStream_t)s, src, lds_, off, dst, ldd, doff, rows, cols, accumulate);
  return LAUNCH_OK();
}
int vt_mish(const float* x, const float* dy, float* out, long n, vt_stream_t s) {
  if (!x || !out || n < 1) return vt_fail(VT_ERR_ARG, "vt_mish: bad argument");
  if (dy) hipLaunchKernelGGL(mish_bwd_kernel, g1(n), dim3(256), 0, (hipStream_t)s, x, dy, out, n);
  else hipLaunchKernelGGL(mish_fwd_kernel, g1(n), dim3(256), 0, (hipStream_t)s, x, out, n);
  return LAUNCH_OK();
}
int vt_gn_mish_bwd(const float* c, const float* gamma, const float* beta, const float* film, const float* dout, float* dc, float* dgamma_part,
                   float* dbeta_part, float* dfilm, int B, int T, int C, int ngroups, float eps, vt_stream_t s) {
  if (!c || !gamma || !beta || !dout || !dc || !dgamma_part || !dbeta_part || (film && !dfilm)) return vt_fail(VT_ERR_ARG, "vt_gn_mish_bwd: null argument");
  if (B < 1 || T < 1 || ngroups < 1 || C % ngroups || C / ngroups > 256) return vt_fail(VT_ERR_ARG, "vt_gn_mish_bwd: bad shape");
  const size_t smem = (size_t)2 * (C / ngroups) * T * sizeof(float);
  if (smem > 64 * 1024) return vt_fail(VT_ERR_UNSUPPORTED, "vt_gn_mish_bwd: group too large for LDS");
  hipLaunchKernelGGL(gn_mish_bwd_kernel, dim3(B * ngroups), dim3(256), smem, (hipStream_t)s, c, gamma, beta, film, dout, dc, dgamma_part, dbeta_part, dfilm, B, T, C,
                     ngroups, eps);
  return LAUNCH_OK();
}
int vt_gelu(const float* x, const float* dy, float* out, long n, vt_stream_t s) {
  if (!x || !out || n < 1) return vt_fail(VT_ERR_ARG, "vt_gelu: bad argument");
  hipLaunchKernelGGL(gelu_kernel, g1(n), dim3(256), 0, (hipStream_t)s, x, dy, out, n);
  return LAUNCH_OK();
}
int vt_si_qsample_ex(const float* x0, const float* x1, const float* z, const float* t, float* xt, float* target_v, float* target_s, float* target_b,
                     float* t_clipped, int B, long per_sample, int gamma_type, float t_min, int interpolant_type, vt_stream_t s) {
  if (!x0 || !x1 || !z || !t || !xt || !target_v || !target_s || !target_b || !t_clipped || B < 1 || per_sample < 1 || gamma_type < 0 || gamma_type > 2 ||
      interpolant_type < 0 || interpolant_type > 6)
    return vt_fail(VT_ERR_ARG, "vt_si_qsample: bad argument");
  hipLaunchKernelGGL(si_qsample_kernel, g1((long)B * per_sample), dim3(256), 0, (hipStream_t)s, x0, x1, z, t, xt, target_v, target_s, target_b, t_clipped, B,
                     per_sample, gamma_type, t_min, interpolant_type);
  return LAUNCH_OK();
}
int vt_si_qsample(const float* x0, const float* x1, const float* z, const float* t, float* xt, float* target_v, float* target_s, float* target_b,
                  float* t_clipped, int B, long per_sample, int gamma_type, float t_min, vt_stream_t s) {
  return vt_si_qsample_ex(x0, x1, z, t, xt, target_v, target_s, target_b, t_clipped, B, per_sample, gamma_type, t_min, 0, s);
}
int vt_si_loss(const float* out, const float* target, float* dout, float* loss, int B, long per_sample, vt_stream_t s) {
  if (!out || !target || !dout || !loss || B < 1 || per_sample < 1) return vt_fail(VT_ERR_ARG, "vt_si_loss: bad argument");
  hipLaunchKernelGGL(si_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, out, target, dout, loss, B, per_sample);
  return LAUNCH_OK();
}
int vt_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, vt_stream_t s) {
  if (!p || !g || !m || !v || n < 1 || step < 1) return vt_fail(VT_ERR_ARG, "vt_adamw: bad argument");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, g1(n), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2));
  return LAUNCH_OK();
}
int vt_train_hyper(float lr, float beta1, float beta2, int step, float ema_decay, float* out4) {   // host: the scalars vt_adamw / vt_ema_update derive
  if (!out4 || step < 1) return vt_fail(VT_ERR_ARG, "vt_train_hyper: bad argument");
  out4[0] = lr;
  out4[1] = 1.0f - powf(beta1, (float)step);
  out4[2] = sqrtf(1.0f - powf(beta2, (float)step));
  out4[3] = 1.0f - ema_decay;
  return VT_OK;
}
int vt_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2, float eps, float weight_decay,
                 vt_stream_t s) {
  if (!p || !g || !m || !v || !hyper || n < 1) return vt_fail(VT_ERR_ARG, "vt_adamw_dev: bad argument");
  hipLaunchKernelGGL(adamw_dev_kernel, g1(n), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, hyper, beta1, beta2, eps, weight_decay);
  return LAUNCH_OK();
}
int vt_adamw_ema_multi(const void* table, int ntensors, long total_chunks, const float* hyper, float beta1, float beta2, float eps, float weight_decay,
                       vt_stream_t s) {
  if (!table || !hyper || ntensors < 1 || total_chunks < 1) return vt_fail(VT_ERR_ARG, "vt_adamw_ema_multi: bad argument");
  hipLaunchKernelGGL(adamw_ema_mt_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)s, (const MtEntry*)table, ntensors, hyper, beta1, beta2,
                     eps, weight_decay);
  return LAUNCH_OK();
}
int vt_ema_update_dev(float* shadow, const float* p, long n, const float* hyper, vt_stream_t s) {
  if (!shadow || !p || !hyper || n < 1) return vt_fail(VT_ERR_ARG, "vt_ema_update_dev: bad argument");
  hipLaunchKernelGGL(ema_dev_kernel, g1(n), dim3(256), 0, (hipStream_t)s, shadow, p, n, hyper);
  return LAUNCH_OK();
}
int vt_ema_update(float* shadow, const float* p, long n, float decay, vt_stream_t s) {
  if (!shadow || !p || n < 1) return vt_fail(VT_ERR_ARG, "vt_ema_update: bad argument");
  hipLaunchKernelGGL(ema_kernel, g1(n), dim3(256), 0, (hipStream_t)s, shadow, p, n, 1.0f - decay);
  return LAUNCH_OK();
}
int vt_posemb(const float* t, float* out, int B, int dim, vt_stream_t s) {
  if (!t || !out || B < 1 || dim < 4 || dim % 2) return vt_fail(VT_ERR_ARG, "vt_posemb: bad argument");
  hipLaunchKernelGGL(posemb_kernel, g1((long)B * dim), dim3(256), 0, (hipStream_t)s, t, out, B, dim);
  return LAUNCH_OK();
}
