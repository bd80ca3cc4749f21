// vt_train_lstm.hip — element-wise pieces of the LSTM residual head's TRAINING step (second head): back-propagation
// through time of `TactileLSTMController.forward` + `get_loss` (residual_controller/lstm_step_controller.py:176-211, 321-337) and the
// AdamW loop of lstm_train.py:26-33, 129-133.  Batch-major sequences [B][T][C] throughout (the reference's batch_first=True): a tick's
// rows are strided by T*C, so every kernel here takes the tick index and addresses the slot itself, and the recurrent GEMM reads /
// writes small contiguous [B][C] scratch matrices.  All matrix products (input projections of all ticks at once, the per-tick
// recurrent product, every weight gradient) go through vt_gemm; what is here is HBM-/launch-bound element-wise work, fp32.
#include <math.h>
#include "vt_common.h"
#include "vt_host.h"
#include "../../include/vlatouch.h"

namespace {

inline dim3 g1(long n) { return dim3((unsigned)((n + 255) / 256)); }
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// One LSTM tick, forward (torch.nn.LSTM gate order i, f, g, o):  a = gx[b][t][:] + gh[b][:]  (gx carries both biases)
//   i = s(a_i) f = s(a_f) g = tanh(a_g) o = s(a_o);  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t)
// saved for the backward pass: act[b][t][4H] = (i,f,g,o), cseq[b][t][H] = c_t; written: hseq[b][t] = h_t, hprev[b][t+1] = h_t
// (hprev[b][t] = h_{t-1}: the left operand of the recurrent weight gradient), hcur[b] = h_t (contiguous, the next tick's GEMM input).
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh, float* __restrict__ act, float* __restrict__ cseq,
                                     float* __restrict__ hseq, float* __restrict__ hprev, float* __restrict__ hcur, int B, int T, int H, int t) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const long row = (long)b * T + t;
  const float* gxr = gx + row * 4 * H;
  const float* ghr = gh ? gh + (long)b * 4 * H : nullptr;
  const float ai = gxr[j] + (ghr ? ghr[j] : 0.f), af = gxr[H + j] + (ghr ? ghr[H + j] : 0.f);
  const float ag = gxr[2 * H + j] + (ghr ? ghr[2 * H + j] : 0.f), ao = gxr[3 * H + j] + (ghr ? ghr[3 * H + j] : 0.f);
  const float i = sigm(ai), f = sigm(af), g = tanhf(ag), o = sigm(ao);
  const float cp = t > 0 ? cseq[(row - 1) * H + j] : 0.f;
  const float c = f * cp + i * g;
  const float h = o * tanhf(c);
  float* ar = act + row * 4 * H;
  ar[j] = i; ar[H + j] = f; ar[2 * H + j] = g; ar[3 * H + j] = o;
  cseq[row * H + j] = c;
  hseq[row * H + j] = h;
  if (t + 1 < T) hprev[(row + 1) * H + j] = h;
  if (t == 0) hprev[row * H + j] = 0.f;
  hcur[idx] = h;
}

// One LSTM tick, backward.  dh = dhseq[b][t] + dh_rec[b] (gradient arriving from tick t+1 through W_hh; null at t = T-1),
// dc = dc_next[b] + dh o (1 - tanh^2 c_t); pre-activation gradients -> dgates[b][t][4H] (all-tick weight / input gradients) and
// dgcur[b][4H] (contiguous: the recurrent data-gradient GEMM's input); dc_next <- dc f.
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dhseq, const float* __restrict__ dh_rec, const float* __restrict__ act,
                                     const float* __restrict__ cseq, float* __restrict__ dc_next, float* __restrict__ dgates, float* __restrict__ dgcur,
                                     int B, int T, int H, int t, int last) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  const long row = (long)b * T + t;
  const float* ar = act + row * 4 * H;
  const float i = ar[j], f = ar[H + j], g = ar[2 * H + j], o = ar[3 * H + j];
  const float tc = tanhf(cseq[row * H + j]);
  const float cp = t > 0 ? cseq[(row - 1) * H + j] : 0.f;
  const float dh = dhseq[row * H + j] + (dh_rec ? dh_rec[idx] : 0.f);
  const float dc = (last ? 0.f : dc_next[idx]) + dh * o * (1.0f - tc * tc);
  const float dai = dc * g * i * (1.0f - i), daf = dc * cp * f * (1.0f - f), dag = dc * i * (1.0f - g * g), dao = dh * tc * o * (1.0f - o);
  dc_next[idx] = dc * f;
  float* dr = dgates + row * 4 * H;
  dr[j] = dai; dr[H + j] = daf; dr[2 * H + j] = dag; dr[3 * H + j] = dao;
  float* dq = dgcur + (long)b * 4 * H;
  dq[j] = dai; dq[H + j] = daf; dq[2 * H + j] = dag; dq[3 * H + j] = dao;
}

// LayerNorm backward per row (one wave per row, two-pass statistics as the forward kernel): xh = (x - mean) rstd,
//   dx = rstd (dy g - mean_c(dy g) - xh mean_c(dy g xh));  dyxh = dy xh (its column sums are d gamma; d beta = column sums of dy)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                     float* __restrict__ dx, float* __restrict__ dyxh, int rows, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * C;
  const float* dr = dy + (long)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)C;
  float v = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
  const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
  float a = 0.f, bsum = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mean) * rstd, dg = dr[c] * gamma[c];
    a += dg; bsum += dg * xh;
  }
  a = wave_sum(a) / (float)C; bsum = wave_sum(bsum) / (float)C;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - mean) * rstd;
    dx[(long)row * C + c] = rstd * (dr[c] * gamma[c] - a - xh * bsum);
    dyxh[(long)row * C + c] = dr[c] * xh;
  }
}

// dst[b][t][doff + c] = src[b][c] for every t (the observation encoding repeated along the sequence, lstm_step_controller.py:199-200)
__global__ void bcast_mid_kernel(const float* __restrict__ src, float* __restrict__ dst, long ldd, int doff, int B, int T, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * T * C) return;
  const int c = (int)(i % C);
  const long row = i / C;
  dst[row * ldd + doff + c] = src[(row / T) * C + c];
}
// out[b][c] = sum_t src[b][t][off + c]  (the gradient of the repeat above)
__global__ void sum_mid_kernel(const float* __restrict__ src, long lds_, int off, float* __restrict__ out, int B, int T, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += src[((long)b * T + t) * lds_ + off + c];
  out[i] = s;
}
__global__ void mul_kernel(float* __restrict__ a, const float* __restrict__ b, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] *= b[i];
}
// pred = base + delta (residual connection, :208-211); F.mse_loss(pred, target) = mean over all elements; d loss / d delta = 2 (pred - target) / n
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ base, const float* __restrict__ delta, const float* __restrict__ target,
                                                  float* __restrict__ pred, float* __restrict__ ddelta, float* __restrict__ loss, long n) {
  __shared__ float red[4];
  float s = 0.f;
  const float k = 2.0f / (float)n;
  for (long i = threadIdx.x; i < n; i += 256) {
    const float p = (base ? base[i] : 0.f) + delta[i], e = p - target[i];
    pred[i] = p;
    ddelta[i] = k * e;
    s += e * e;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

}  // namespace

extern "C" {

int vt_lstm_cell_fwd(const float* gx, const float* gh, float* act, float* cseq, float* hseq, float* hprev, float* hcur, int B, int T, int H, int t,
                     vt_stream_t s) {
  if (!gx || !act || !cseq || !hseq || !hprev || !hcur || B < 1 || T < 1 || H < 1 || t < 0 || t >= T || (t > 0 && !gh))
    return vt_fail(VT_ERR_ARG, "vt_lstm_cell_fwd: bad argument");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, g1((long)B * H), dim3(256), 0, (hipStream_t)s, gx, gh, act, cseq, hseq, hprev, hcur, B, T, H, t);
  return vt_check_launch();
}
int vt_lstm_cell_bwd(const float* dhseq, const float* dh_rec, const float* act, const float* cseq, float* dc_next, float* dgates, float* dgcur,
                     int B, int T, int H, int t, vt_stream_t s) {
  if (!dhseq || !act || !cseq || !dc_next || !dgates || !dgcur || B < 1 || T < 1 || H < 1 || t < 0 || t >= T || (t < T - 1 && !dh_rec))
    return vt_fail(VT_ERR_ARG, "vt_lstm_cell_bwd: bad argument");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, g1((long)B * H), dim3(256), 0, (hipStream_t)s, dhseq, dh_rec, act, cseq, dc_next, dgates, dgcur, B, T, H, t,
                     t == T - 1 ? 1 : 0);
  return vt_check_launch();
}
int vt_ln_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dyxh, int rows, int C, float eps, vt_stream_t s) {
  if (!x || !gamma || !dy || !dx || !dyxh || rows < 1 || C < 1) return vt_fail(VT_ERR_ARG, "vt_ln_bwd: bad argument");
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)s, x, gamma, dy, dx, dyxh, rows, C, eps);
  return vt_check_launch();
}
int vt_bcast_mid(const float* src, float* dst, long ldd, int doff, int B, int T, int C, vt_stream_t s) {
  if (!src || !dst || B < 1 || T < 1 || C < 1 || doff < 0 || ldd < doff + C) return vt_fail(VT_ERR_ARG, "vt_bcast_mid: bad argument");
  hipLaunchKernelGGL(bcast_mid_kernel, g1((long)B * T * C), dim3(256), 0, (hipStream_t)s, src, dst, ldd, doff, B, T, C);
  return vt_check_launch();
}
int vt_sum_mid(const float* src, long lds_, int off, float* out, int B, int T, int C, vt_stream_t s) {
  if (!src || !out || B < 1 || T < 1 || C < 1 || off < 0 || lds_ < off + C) return vt_fail(VT_ERR_ARG, "vt_sum_mid: bad argument");
  hipLaunchKernelGGL(sum_mid_kernel, g1((long)B * C), dim3(256), 0, (hipStream_t)s, src, lds_, off, out, B, T, C);
  return vt_check_launch();
}
int vt_mul_(float* a, const float* b, long n, vt_stream_t s) {
  if (!a || !b || n < 1) return vt_fail(VT_ERR_ARG, "vt_mul_: bad argument");
  hipLaunchKernelGGL(mul_kernel, g1(n), dim3(256), 0, (hipStream_t)s, a, b, n);
  return vt_check_launch();
}
int vt_mse_residual(const float* base, const float* delta, const float* target, float* pred, float* ddelta, float* loss, long n, vt_stream_t s) {
  if (!delta || !target || !pred || !ddelta || !loss || n < 1) return vt_fail(VT_ERR_ARG, "vt_mse_residual: bad argument");
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, base, delta, target, pred, ddelta, loss, n);
  return vt_check_launch();
}

}  // extern "C"
