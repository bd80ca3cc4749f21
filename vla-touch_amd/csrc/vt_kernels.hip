// vt_kernels.hip — memory-bound / small kernels of the refinement path (gfx950).
// Row-wise norms (LayerNorm / RMSNorm, one wave per row, 16-B loads), GroupNorm+Mish+FiLM+residual
// over split-K partial slabs (one wave per (sample, group), group staged in LDS), per-head q/k RMSNorm,
// image statistics + patchify (one HBM read of the frames), the SDE update, action (de)normalisation
// and the small glue kernels of the U-Net / LSTM drivers.
#include <stdlib.h>
#include <type_traits>
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

// ------------------------------------------------------------------ row norm (LayerNorm / RMSNorm)
// one wave per row; D % 4 == 0; D <= 64*4*MAXV
template <typename TI, typename TO, int MAXV>
__global__ __launch_bounds__(256) void rownorm_kernel(const TI* __restrict__ x, long ldx, TO* __restrict__ y, long ldy,
                                                      const float* __restrict__ w, const float* __restrict__ b,
                                                      int rows, int D, float eps, int mode, unsigned* range_flag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* xr = x + (long)row * ldx;
  float v[MAXV][4];
  const int nv = D >> 2;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][j] = ldf<TI>(xr, (size_t)c4 * 4 + j); s += v[i][j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = 0.f;
    }
  }
  float mean = 0.f, var;
  if (mode == VT_NORM_RMS_MEANSQ) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) q += v[i][j] * v[i][j];
    var = wave_sum(q) / (float)D;
  } else {
    mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
      }
    }
    q = wave_sum(q);
    var = (mode == VT_NORM_RMS_VAR) ? q / (float)(D - 1) : q / (float)D;
    if (mode == VT_NORM_RMS_VAR) mean = 0.f;   // timm<=1.0.8 rms_norm: x * rsqrt(var_unbiased(x) + eps), x not centred
  }
  if (range_flag && lane == 0 && vt_nonfinite(var)) vt_range_note(range_flag, VT_RANGE_NONFINITE);
  const float rstd = rsqrtf(var + eps);
  TO* yr = y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c4 * 4 + j;
        float o = (v[i][j] - mean) * rstd * w[c];
        if (b) o += b[c];
        stf<TO>(yr, c, o);
      }
    }
  }
}

// wide rows (D >= 1024, fp32 in): one 256-thread block per row, 16-B loads, two-level (wave, LDS) reductions — 4x the
// blocks of the wave-per-row kernel for the [B*67, 2048] RMSNorms of the RDT step loop
template <typename TO, int NV>
__global__ __launch_bounds__(256) void rownorm_block_kernel(const float* __restrict__ x, long ldx, TO* __restrict__ y, long ldy, const float* __restrict__ w,
                                                            const float* __restrict__ b, int D, float eps, int mode, unsigned* range_flag) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* xr = x + (long)blockIdx.x * ldx;
  const int nv = D >> 2;
  float4 v[NV];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = tid + 256 * i;
    v[i] = c4 < nv ? *reinterpret_cast<const float4*>(xr + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  auto block_sum = [&](float t) {
    t = wave_sum(t);
    __syncthreads();
    if (lane == 0) red[wv] = t;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  float mean = 0.f, var;
  if (mode == VT_NORM_RMS_MEANSQ) {
    var = block_sum(q) / (float)D;
  } else {
    mean = block_sum(s) / (float)D;
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (tid + 256 * i < nv) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        d2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    d2 = block_sum(d2);
    var = (mode == VT_NORM_RMS_VAR) ? d2 / (float)(D - 1) : d2 / (float)D;
    if (mode == VT_NORM_RMS_VAR) mean = 0.f;
  }
  if (range_flag && tid == 0 && vt_nonfinite(var)) vt_range_note(range_flag, VT_RANGE_NONFINITE);     // an inf / NaN anywhere in the row makes its statistics non-finite
  const float rstd = rsqrtf(var + eps);
  TO* yr = y + (long)blockIdx.x * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = tid + 256 * i;
    if (c4 < nv) {
      const float4 ww = *reinterpret_cast<const float4*>(w + c4 * 4);
      float o[4] = {(v[i].x - mean) * rstd * ww.x, (v[i].y - mean) * rstd * ww.y, (v[i].z - mean) * rstd * ww.z, (v[i].w - mean) * rstd * ww.w};
      if (b) { const float4 bb = *reinterpret_cast<const float4*>(b + c4 * 4); o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w; }
      if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(yr + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
      else {
        TO t[4] = {Elem<TO>::from_f(o[0]), Elem<TO>::from_f(o[1]), Elem<TO>::from_f(o[2]), Elem<TO>::from_f(o[3])};
        *reinterpret_cast<uint2*>(yr + c4 * 4) = *reinterpret_cast<const uint2*>(t);
      }
    }
  }
}

// many wide rows (the ViT towers: 139 968 x 1152 for the SigLIP step, 16 448 x 768 for DINOv2): one WAVE per row, 4 rows per block, 16-byte
// loads held in registers, reductions by wave shuffles only — no LDS, no block barrier (the block-per-row kernel above pays two to four
// __syncthreads() per row and keeps a quarter of the bytes in flight per CU; it stays for the few-row RMSNorms of the RDT step loop)
template <typename TO, int NV>
__global__ __launch_bounds__(256) void rownorm_wave_kernel(const float* __restrict__ x, long ldx, TO* __restrict__ y, long ldy, const float* __restrict__ w,
                                                           const float* __restrict__ b, int rows, int D, float eps, int mode, unsigned* range_flag) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  const int nv = D >> 2;
  float4 v[NV];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    v[i] = c4 < nv ? *reinterpret_cast<const float4*>(xr + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  float mean = 0.f, var;
  if (mode == VT_NORM_RMS_MEANSQ) {
    var = wave_sum(q) / (float)D;
  } else {
    mean = wave_sum(s) / (float)D;
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < nv) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        d2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    d2 = wave_sum(d2);
    var = (mode == VT_NORM_RMS_VAR) ? d2 / (float)(D - 1) : d2 / (float)D;
    if (mode == VT_NORM_RMS_VAR) mean = 0.f;
  }
  if (range_flag && lane == 0 && vt_nonfinite(var)) vt_range_note(range_flag, VT_RANGE_NONFINITE);
  const float rstd = rsqrtf(var + eps);
  TO* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 ww = *reinterpret_cast<const float4*>(w + c4 * 4);
      float o[4] = {(v[i].x - mean) * rstd * ww.x, (v[i].y - mean) * rstd * ww.y, (v[i].z - mean) * rstd * ww.z, (v[i].w - mean) * rstd * ww.w};
      if (b) { const float4 bb = *reinterpret_cast<const float4*>(b + c4 * 4); o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w; }
      if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(yr + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
      else {
        TO t[4] = {Elem<TO>::from_f(o[0]), Elem<TO>::from_f(o[1]), Elem<TO>::from_f(o[2]), Elem<TO>::from_f(o[3])};
        *reinterpret_cast<uint2*>(yr + c4 * 4) = *reinterpret_cast<const uint2*>(t);
      }
    }
  }
}

// ------------------------------------------------------------------ per-head RMSNorm on 64-wide rows (q / k norm)
template <typename T>
__global__ __launch_bounds__(256) void headnorm_kernel(T* __restrict__ x, long tok_stride, int heads, long rows,
                                                       const float* __restrict__ w, float eps, int mode) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const long tok = r / heads;
  const int h = (int)(r - tok * heads);
  T* p = x + tok * tok_stride + (long)h * 64;
  const float v = ldf<T>(p, lane);
  float var;
  if (mode == VT_NORM_RMS_VAR) {
    const float mean = wave_sum(v) * (1.f / 64.f);
    const float d = v - mean;
    var = wave_sum(d * d) * (1.f / 63.f);
  } else {
    var = wave_sum(v * v) * (1.f / 64.f);
  }
  stf<T>(p, lane, v * rsqrtf(var + eps) * w[lane]);
}

// ------------------------------------------------------------------ GroupNorm + Mish (+FiLM) (+residual) over partial slabs
// One 256-thread block per (net, sample, group): the unit has only C/8 * T = 256..1024 values but each is the sum of up to 8
// split-K slabs, so the kernel is a latency chain of small loads; a block per unit (instead of a wave) quarters the chain and
// the slab loads of a value are issued together (unrolled, predicated) rather than one dependent add at a time.
constexpr int GN_MAX_SLABS = 8;
template <typename TO>
__global__ __launch_bounds__(256) void gn_kernel(const VtGnParams p) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int unit = blockIdx.x;
  const int cpg = p.C / p.ngroups;
  const int n = cpg * p.T;
  const int net = unit / (p.B * p.ngroups);
  const int rem = unit - net * p.B * p.ngroups;
  const int b = rem / p.ngroups, grp = rem - b * p.ngroups;
  const float* P = p.P + (long)net * p.p_gs;
  const float* bias = p.bias ? p.bias + (long)net * p.vec_gs : nullptr;
  const float* gamma = p.gamma + (long)net * p.vec_gs;
  const float* beta = p.beta + (long)net * p.vec_gs;
  const int c0 = grp * cpg;
  auto block_sum = [&](float t) {
    t = wave_sum(t);
    __syncthreads();
    if (lane == 0) red[wv] = t;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  float s = 0.f;
  for (int e = tid; e < n; e += 256) {
    const int t = e / cpg, c = e - t * cpg;
    const long off = ((long)b * p.T + t) * p.ldp + c0 + c;
    float part[GN_MAX_SLABS];
#pragma unroll
    for (int k = 0; k < GN_MAX_SLABS; ++k) part[k] = k < p.nslabs ? P[(long)k * p.slab_stride + off] : 0.f;
    float v = bias ? bias[c0 + c] : 0.f;
#pragma unroll
    for (int k = 0; k < GN_MAX_SLABS; ++k) v += part[k];          // same order as a sequential sum over the slabs
    for (int k = GN_MAX_SLABS; k < p.nslabs; ++k) v += P[(long)k * p.slab_stride + off];
    gsm[e] = v;
    s += v;
  }
  const float mean = block_sum(s) / (float)n;
  float q = 0.f;
  for (int e = tid; e < n; e += 256) { const float d = gsm[e] - mean; q += d * d; }
  const float rstd = rsqrtf(block_sum(q) / (float)n + p.eps);
  const float* film = p.film ? p.film + (long)net * p.film_gs + (long)b * p.film_ld + p.film_off : nullptr;
  const TO* R = p.residual ? reinterpret_cast<const TO*>(p.residual) + (long)net * p.r_gs : nullptr;
  TO* O = reinterpret_cast<TO*>(p.out) + (long)net * p.o_gs;
  for (int e = tid; e < n; e += 256) {
    const int t = e / cpg, c = e - t * cpg, col = c0 + c;
    float y = (gsm[e] - mean) * rstd * gamma[col] + beta[col];
    y = act_apply(y, VT_ACT_MISH);
    if (film) y = film[col] * y + film[p.C + col];
    const long row = (long)b * p.T + t;
    if (R) y += Elem<TO>::to_f(R[row * p.ldr + col]);
    O[row * p.ldo + col] = Elem<TO>::from_f(y);
  }
}

// Prefetch-only blocks of the slab-reduction launches (small-batch RDT path): blocks pf_block0 .. of the grid touch one dword per 64 bytes of the NEXT
// GEMM's frozen weights and exit — the weights stream HBM -> Infinity Cache on CUs the reduction leaves idle (M = 67 rows use a quarter of the chip),
// instead of at the head of the next launch.  Fire and forget: the hardware retires the loads before s_endpgm.
constexpr int VT_PF_BLOCKS = 128;
__device__ __forceinline__ void vt_prefetch_block(const void* pf_ptr, unsigned pf_bytes, int pb, int tid) {
  unsigned tmp = 0;
  for (unsigned off = (unsigned)(pb * 256 + tid) * 64u; off < pf_bytes; off += (unsigned)VT_PF_BLOCKS * 256u * 64u)
    asm volatile("global_load_dword %0, %1, off" : "=v"(tmp) : "v"(reinterpret_cast<const char*>(pf_ptr) + off) : "memory");
  asm volatile("" :: "v"(tmp));
}

// ------------------------------------------------------------------ split-K slab reduction + Linear epilogue (small-M GEMMs)
// out[m, n..n+3] = residual + colscale * act(sum_s slab[s][m][n] + bias): the tail of a GEMM whose k range was split over
// blocks because M alone gives too few tiles to hide a 2048-deep k-loop (RDT at batch 1-4: M = 67..268 rows).
// Optional per-head RMSNorm (q_norm / k_norm): a head's 64 columns are the 4 columns of 16 consecutive threads (N % 64 == 0), reduced
// with DPP; columns [0, hn_c0) use gains w0, [hn_c0, hn_c1) use w1.
template <typename TO>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int S, long slab_stride, int M, int N, const float* __restrict__ bias,
                                                          int act, const float* __restrict__ cs, const TO* __restrict__ R, long ldr, TO* __restrict__ out, long ldo,
                                                          const float* __restrict__ hn_w0, const float* __restrict__ hn_w1, int hn_c0, int hn_c1, float hn_eps, int hn_mode,
                                                          const void* pf_ptr, unsigned pf_bytes, int pf_block0) {
  if (pf_ptr && (int)blockIdx.x >= pf_block0) { vt_prefetch_block(pf_ptr, pf_bytes, blockIdx.x - pf_block0, threadIdx.x); return; }
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  const bool live = i < (long)M * n4;
  if (!live && !hn_w0) return;
  const long ii = live ? i : (long)M * n4 - 1;         // with a head norm every lane takes part in the DPP reduction
  const int m = (int)(ii / n4), n = (int)(ii - (long)m * n4) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < S; s0 += 4) {            // 4 independent loads in flight
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = s0 + u < S ? *reinterpret_cast<const float4*>(slabs + (long)(s0 + u) * slab_stride + (long)m * N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  float o[4] = {acc.x, acc.y, acc.z, acc.w};
  if (hn_w0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += bias ? bias[n + r] : 0.f;
    const float sm = row16_sum((o[0] + o[1]) + (o[2] + o[3]));
    const float sq = row16_sum((o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]));
    const float* hw = n < hn_c0 ? hn_w0 : (hn_w1 && n < hn_c1 ? hn_w1 : nullptr);
    if (hw) {
      float var;
      if (hn_mode == 2) { const float mean = sm * (1.f / 64.f); var = (sq - 64.f * mean * mean) * (1.f / 63.f); }
      else var = sq * (1.f / 64.f);
      const float rstd = rsqrtf(var + hn_eps);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] *= rstd * hw[(n & 63) + r];
    }
    if (!live) return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = o[r] + ((bias && !hn_w0) ? bias[n + r] : 0.f);
    x = act_apply(x, act);
    if (cs) x *= cs[n + r];
    if (R) x += Elem<TO>::to_f(R[(long)m * ldr + n + r]);
    out[(long)m * ldo + n + r] = Elem<TO>::from_f(x);
  }
}

// Residual Linear + the norm that follows it, for the split-K path: x[m, :] += sum_s slab[s][m, :] + bias (fp32 stream, in place) and
// xn[m, :] = norm(x[m, :]) * w (+ b) in the activation dtype — one block per row, so the next Linear needs no separate norm launch.
template <typename TN, int NV>
__global__ __launch_bounds__(256) void slab_reduce_norm_kernel(const float* __restrict__ slabs, int S, long slab_stride, int N, const float* __restrict__ bias,
                                                               float* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ b,
                                                               float eps, int mode, TN* __restrict__ xn, long ldxn, const void* pf_ptr, unsigned pf_bytes, int pf_block0) {
  if (pf_ptr && (int)blockIdx.x >= pf_block0) { vt_prefetch_block(pf_ptr, pf_bytes, blockIdx.x - pf_block0, threadIdx.x); return; }
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, m = blockIdx.x;
  const int nv = N >> 2;
  float4 v[NV];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = tid + 256 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < nv) {
      float4 acc = *reinterpret_cast<const float4*>(x + (long)m * ldx + c4 * 4);
      if (bias) { const float4 bb = *reinterpret_cast<const float4*>(bias + c4 * 4); acc.x += bb.x; acc.y += bb.y; acc.z += bb.z; acc.w += bb.w; }
      for (int k0 = 0; k0 < S; k0 += 4) {          // 4 independent loads in flight
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          t[u] = k0 + u < S ? *reinterpret_cast<const float4*>(slabs + (long)(k0 + u) * slab_stride + (long)m * N + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
      }
      *reinterpret_cast<float4*>(x + (long)m * ldx + c4 * 4) = acc;
      v[i] = acc;
      s += (acc.x + acc.y) + (acc.z + acc.w);
      q += (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
    }
  }
  auto block_sum = [&](float t) {
    t = wave_sum(t);
    __syncthreads();
    if (lane == 0) red[wv] = t;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  float mean = 0.f, var;
  if (mode == VT_NORM_RMS_MEANSQ) {
    var = block_sum(q) / (float)N;
  } else {
    mean = block_sum(s) / (float)N;
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (tid + 256 * i < nv) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        d2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    d2 = block_sum(d2);
    var = (mode == VT_NORM_RMS_VAR) ? d2 / (float)(N - 1) : d2 / (float)N;
    if (mode == VT_NORM_RMS_VAR) mean = 0.f;
  }
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = tid + 256 * i;
    if (c4 < nv) {
      const float4 ww = *reinterpret_cast<const float4*>(w + c4 * 4);
      float o[4] = {(v[i].x - mean) * rstd * ww.x, (v[i].y - mean) * rstd * ww.y, (v[i].z - mean) * rstd * ww.z, (v[i].w - mean) * rstd * ww.w};
      if (b) { const float4 bb = *reinterpret_cast<const float4*>(b + c4 * 4); o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w; }
      TN* yr = xn + (long)m * ldxn;
      if constexpr (sizeof(TN) == 4) *reinterpret_cast<float4*>(yr + c4 * 4) = make_float4(o[0], o[1], o[2], o[3]);
      else {
        TN t[4] = {Elem<TN>::from_f(o[0]), Elem<TN>::from_f(o[1]), Elem<TN>::from_f(o[2]), Elem<TN>::from_f(o[3])};
        *reinterpret_cast<uint2*>(yr + c4 * 4) = *reinterpret_cast<const uint2*>(t);
      }
    }
  }
}

// ------------------------------------------------------------------ small element-wise kernels
template <typename TO>
__global__ void sinusoid_kernel(const float* __restrict__ t, float t_host, TO* __restrict__ out, int B, int dim, int nets, long net_stride, int cos_first, float denom_minus) {
  // out[net][b][i]: U-Net style (sin | cos, freq = exp(-i*ln(1e4)/(half-1)))  or  RDT style (cos | sin, exp(-i*ln(1e4)/half))
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim >> 1;
  if (i >= B * dim) return;
  const int b = i / dim, c = i - b * dim;
  const int k = c < half ? c : c - half;
  // U-Net: exp(k * -(ln 1e4 / (half-1))) (conditional_unet_1D.py:15-16); RDT: exp(-ln 1e4 * k / half) (blocks.py:53-56)
  const float f = cos_first ? expf(-9.210340371976184f * (float)k / (float)half)
                            : expf((float)k * -(9.210340371976184f / ((float)half - denom_minus)));
  const float a = (t ? t[b] : t_host) * f;
  const bool first = c < half;
  const float v = (first == (cos_first != 0)) ? cosf(a) : sinf(a);
  for (int n = 0; n < nets; ++n) out[(long)n * net_stride + i] = Elem<TO>::from_f(v);
}

template <typename TI, typename TO>
__global__ void act_copy_kernel(const TI* __restrict__ in, long ldi, TO* __restrict__ out, long ldo, int rows, int cols, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
  out[(long)r * ldo + c] = Elem<TO>::from_f(act_apply(Elem<TI>::to_f(in[(long)r * ldi + c]), act));
}

// SwiGLU gate, in place: h[r][c] = silu(h[r][c]) * h[r][F + c] for c < F (rows of 2 F values; HF Dinov2SwiGLUFFN: x1, x2 = hidden.chunk(2)); 8 / 4 elements per thread
template <typename T>
__global__ void swiglu_kernel(T* __restrict__ h, long ld, long rows, int F, unsigned* range_flag) {
  constexpr int V = 16 / sizeof(T);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = F / V;
  if (i >= rows * per_row) return;
  const long r = i / per_row;
  const int c = (int)(i - r * per_row) * V;
  T* p1 = h + r * ld + c;
  uint4 a = *reinterpret_cast<const uint4*>(p1), b = *reinterpret_cast<const uint4*>(p1 + F);
  T* av = reinterpret_cast<T*>(&a);
  const T* bv = reinterpret_cast<const T*>(&b);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const float x1 = Elem<T>::to_f(av[k]), x2 = Elem<T>::to_f(bv[k]);
    float g = x1 * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x1)) * x2;
    // IEEE fp16 storage (the low-precision DINOv2 mode): the gated product of two fp16 values can leave the fp16 range (outlier tokens of real giant
    // checkpoints); saturate instead of producing inf, which fc2 would turn into NaN for the whole row (inf - inf across the k range)
    if constexpr (sizeof(T) == 2 && !std::is_same<T, bf16_t>::value) {
      if (fabsf(g) > 65504.0f) vt_range_note(range_flag, VT_RANGE_GATE_SAT);      // the clamp is not silent (vt_dino_set_range_flag)
      g = fminf(fmaxf(g, -65504.0f), 65504.0f);
    }
    av[k] = Elem<T>::from_f(g);
  }
  *reinterpret_cast<uint4*>(p1) = a;
}

__global__ void sde_update_kernel(float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ s,
                                  const float* __restrict__ z, long n, float dt, float gi, float gdg, float eps, float noise_scale, float d,
                                  float score_eps, int backward) {
  // bridge_model.py:363-385 in the reference's operation order; eps = epsilon(t) of the b term (:369, also when direction='backward'),
  // score_eps = score_weight * epsilon(t or 1 - t) (:376, :380)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sv = s[i] * gi;
  const float b = v[i] - gdg * sv * eps;
  float xn = backward ? x[i] - (b - score_eps * sv) * dt : x[i] + (b + score_eps * sv) * dt;
  if (z) xn += noise_scale * (d * z[i]);
  x[i] = xn;
}

__global__ void actnorm_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ mins,
                               const float* __restrict__ maxs, long n, int dim, float pad, int denorm) {
  // controller_dataset.py:303-346 (normalise: <1e-6 range guard) / :349-384 (denormalise: no guard)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % dim);
  const float mn = mins[c], mx = maxs[c];
  const float pr = (mx - mn) * pad;
  const float ctr = (mn + mx) / 2;
  const float pmin = ctr - pr / 2, pmax = ctr + pr / 2;
  float rng = pmax - pmin;
  if (denorm) {
    out[i] = (in[i] + 1.0f) / 2.0f * rng + pmin;
  } else {
    if (rng < 1e-6f) rng = 1.0f;
    out[i] = 2.0f * (in[i] - pmin) / rng - 1.0f;
  }
}

template <typename TO>
__global__ void pad_cols_kernel(const float* __restrict__ in, int cin, TO* __restrict__ out, int cout, long rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cout) return;
  const long r = i / cout;
  const int c = (int)(i - r * cout);
  out[i] = Elem<TO>::from_f(c < cin ? in[r * cin + c] : 0.f);
}

// out[r][off : off+cols] = (T) src[r][:]   (concat builder, zero padding is the caller's memset)
template <typename TI, typename TO>
__global__ void place_cols_kernel(const TI* __restrict__ src, long lds_, TO* __restrict__ out, long ldo, int off, int rows, int cols) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
  out[(long)r * ldo + off + c] = Elem<TO>::from_f(Elem<TI>::to_f(src[(long)r * lds_ + c]));
}

// rows[b*ld .. ] = vec (broadcast a D-vector to one row per sample: the CLS token + its position embedding)
__global__ void bcast_row_kernel(const float* __restrict__ vec, float* __restrict__ out, long row_stride, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, c = i - b * D;
  out[(long)b * row_stride + c] = vec[c];
}

// ------------------------------------------------------------------ image statistics (max, sum) -> branch flags
template <typename TI>
__global__ __launch_bounds__(256) void imgstat_partial_kernel(const TI* __restrict__ img, long n, float* __restrict__ part) {
  // 16 bytes per lane and four independent loads in flight (a grid-stride loop of 4-byte loads was 73 dependent round trips: 31 us for 19 MB)
  constexpr int V = 16 / sizeof(TI);                       // elements per 16-byte vector
  float mx = -3.4e38f, sm = 0.f;
  const long nvec = ((reinterpret_cast<size_t>(img) & 15) == 0) ? n / V : 0;
  const long stride = (long)gridDim.x * 256;
  auto take = [&](const uint4 q) {
    if constexpr (sizeof(TI) == 4) {
      const float* f = reinterpret_cast<const float*>(&q);
      mx = fmaxf(fmaxf(mx, fmaxf(f[0], f[1])), fmaxf(f[2], f[3]));
      sm += (f[0] + f[1]) + (f[2] + f[3]);
    } else {
      const uint8_t* u = reinterpret_cast<const uint8_t*>(&q);
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) { const float v = (float)u[k]; mx = fmaxf(mx, v); t += v; }
      sm += t;
    }
  };
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const uint4* v4 = reinterpret_cast<const uint4*>(img);
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    const uint4 q0 = v4[i], q1 = v4[i + stride], q2 = v4[i + 2 * stride], q3 = v4[i + 3 * stride];
    take(q0); take(q1); take(q2); take(q3);
  }
  for (; i < nvec; i += stride) take(v4[i]);
  for (long e = nvec * V + (long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {      // unaligned base or a tail that is not a whole vector
    const float v = (float)img[e];
    mx = fmaxf(mx, v);
    sm += v;
  }
  __shared__ float smx[4], ssm[4];
  mx = wave_max(mx);
  sm = wave_sum(sm);
  if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; ssm[threadIdx.x >> 6] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    part[2 * blockIdx.x + 1] = (ssm[0] + ssm[1]) + (ssm[2] + ssm[3]);
  }
}
// flags[0] = pixel scale (1 or 1/255), flags[1] = 1.0 if ImageNet normalisation applies (visual_encoder.py:78,100).  One wave: lane l folds partials l, l + 64, ...
// (a single lane walking all of them was 19 us)
__global__ __launch_bounds__(64) void imgstat_final_kernel(const float* __restrict__ part, int nparts, long n, float pre_scale, int norm_mode, float* __restrict__ flags,
                                                           float* __restrict__ flags_copy) {
  if (blockIdx.x != 0) return;
  float mx = -3.4e38f;
  double sm = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) { mx = fmaxf(mx, part[2 * i]); sm += (double)part[2 * i + 1]; }
  mx = wave_max(mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long bits = __builtin_bit_cast(long long, sm);
    const int lo = __shfl_xor((int)(bits & 0xffffffffll), o, 64), hi = __shfl_xor((int)(bits >> 32), o, 64);
    sm += __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
  }
  if (threadIdx.x != 0) return;
  mx *= pre_scale;
  float scale = pre_scale;
  if (mx > 1.0f) scale = pre_scale / 255.0f;
  const float mean = (float)(sm / (double)n) * scale;
  flags[0] = scale;
  flags[1] = norm_mode == VT_IMGNORM_AUTO ? (mean < 0.5f ? 0.f : 1.f) : (norm_mode == VT_IMGNORM_ON ? 1.f : 0.f);
  flags[2] = mx;
  flags[3] = mean;
  if (flags_copy) { flags_copy[0] = flags[0]; flags_copy[1] = flags[1]; flags_copy[2] = mx; flags_copy[3] = mean; }     // the caller's copy (no runtime copy kernel)
}

// ------------------------------------------------------------------ patchify: frames -> A[b*np + p][c*196 + i*14 + j] (K padded)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const TI* __restrict__ img, int nhwc, int B, int res, int grid_, int kpad,
                                                       const float* __restrict__ flags, TO* __restrict__ out) {
  // one block per (image, patch row): reads 14 full image rows per channel -> coalesced along W
  const int b = blockIdx.x / grid_, py = blockIdx.x - b * grid_;
  const float scale = flags[0];
  const bool norm = flags[1] != 0.f;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
  const int W14 = grid_ * 14;   // used width (res may exceed grid*14; HF conv drops the remainder)
  const int total = 3 * 14 * W14;
  if (!nhwc && sizeof(TI) == 4 && sizeof(TO) == 2 && (res & 1) == 0 && (kpad & 1) == 0 && (reinterpret_cast<size_t>(img) & 7) == 0) {
    // planar fp32 frames -> 16-bit patches, two pixels per lane: x and x + 1 (x even) sit in the same patch (14 is even), so one 8-byte load and one 4-byte
    // store replace two 4-byte loads and two 2-byte stores (the one-pixel loop below: 29 us per 32 frames of 224 x 224)
    const int half = total >> 1, Wh = W14 >> 1;
    for (int e = threadIdx.x; e < half; e += 256) {
      const int c = e / (14 * Wh), r = e - c * 14 * Wh, i = r / Wh, xcol = (r - i * Wh) * 2;
      const int yrow = py * 14 + i;
      const float2 p2 = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(img) + (((long)b * 3 + c) * res + yrow) * res + xcol);
      float v0 = p2.x * scale, v1 = p2.y * scale;
      if (norm) { v0 = (v0 - mean[c]) * istd[c]; v1 = (v1 - mean[c]) * istd[c]; }
      const int px = xcol / 14, j = xcol - px * 14;
      TO o2[2] = {Elem<TO>::from_f(v0), Elem<TO>::from_f(v1)};
      *reinterpret_cast<uint32_t*>(out + ((long)(b * grid_ + py) * grid_ + px) * kpad + c * 196 + i * 14 + j) = *reinterpret_cast<const uint32_t*>(o2);
    }
  } else
  for (int e = threadIdx.x; e < total; e += 256) {
    int c, i, xcol;
    if (nhwc) { i = e / (W14 * 3); const int r = e - i * W14 * 3; xcol = r / 3; c = r - xcol * 3; }
    else      { c = e / (14 * W14); const int r = e - c * 14 * W14; i = r / W14; xcol = r - i * W14; }
    const int yrow = py * 14 + i;
    const long src = nhwc ? (((long)b * res + yrow) * res + xcol) * 3 + c : (((long)b * 3 + c) * res + yrow) * res + xcol;
    float v = (float)img[src] * scale;
    if (norm) v = (v - mean[c]) * istd[c];
    const int px = xcol / 14, j = xcol - px * 14;
    out[((long)(b * grid_ + py) * grid_ + px) * kpad + c * 196 + i * 14 + j] = Elem<TO>::from_f(v);
  }
  // zero the K padding of this patch row
  const int padn = kpad - 588;
  for (int e = threadIdx.x; e < grid_ * padn; e += 256) {
    const int px = e / padn, k = e - px * padn;
    out[((long)(b * grid_ + py) * grid_ + px) * kpad + 588 + k] = Elem<TO>::from_f(0.f);
  }
}

// ------------------------------------------------------------------ LSTM cell (gates pre-computed by two GEMMs)
__global__ void lstm_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h, float* __restrict__ c, int B, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * H) return;
  const int b = i / H, j = i - b * H;
  const float* a = gi + (long)b * 4 * H;
  const float* r = gh + (long)b * 4 * H;
  const float ig = 1.f / (1.f + expf(-(a[j] + r[j])));
  const float fg = 1.f / (1.f + expf(-(a[H + j] + r[H + j])));
  const float gg = tanhf(a[2 * H + j] + r[2 * H + j]);
  const float og = 1.f / (1.f + expf(-(a[3 * H + j] + r[3 * H + j])));
  const float cn = fg * c[i] + ig * gg;
  c[i] = cn;
  h[i] = og * tanhf(cn);
}

// DPM-Solver++ step on [n] elements: x = a*x + b0*m0 + b1*m1  (x fp32 master, xb = bf16/f32 copy for the next forward)
template <typename TM>
__global__ void axpby3_kernel(float* __restrict__ x, const TM* __restrict__ m0, const TM* __restrict__ m1, float a, float b0, float b1, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = a * x[i] + b0 * Elem<TM>::to_f(m0[i]);
  if (m1) v += b1 * Elem<TM>::to_f(m1[i]);
  x[i] = v;
}

inline dim3 g1(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

// =============================================================================== host launchers
#define DISPATCH_T(dt, T, ...) \
  if ((dt) == VT_F32) { using T = float; __VA_ARGS__; } else if ((dt) == VT_F16) { using T = half_t; __VA_ARGS__; } else { using T = bf16_t; __VA_ARGS__; }

int vt_k_rownorm(const void* x, int xdt, long ldx, void* y, int ydt, long ldy, const float* w, const float* b, int rows, int D,
                 float eps, int mode, hipStream_t s, unsigned* range_flag) {
  if (D % 4 || D > 64 * 4 * 8 || rows <= 0) return VT_ERR_ARG;
  static const int wave_rows = [] { const char* e = getenv("VLATOUCH_ROWNORM_WAVE"); return e ? atoi(e) : 8192; }();   // rows from which the wave-per-row kernel takes over (0 = never)
  if (xdt == VT_F32 && D >= 512 && wave_rows > 0 && rows >= wave_rows && (ldx % 4) == 0 && (ldy % 4) == 0) {
    const dim3 grid((unsigned)((rows + 3) / 4));
#define VT_RNW(TO, NV) hipLaunchKernelGGL((rownorm_wave_kernel<TO, NV>), grid, dim3(256), 0, s, (const float*)x, ldx, (TO*)y, ldy, w, b, rows, D, eps, mode, range_flag)
#define VT_RNW_T(TO) do { if (D <= 1024) VT_RNW(TO, 4); else if (D <= 1280) VT_RNW(TO, 5); else VT_RNW(TO, 8); } while (0)
    if (ydt == VT_F32) VT_RNW_T(float); else if (ydt == VT_F16) VT_RNW_T(half_t); else VT_RNW_T(bf16_t);
#undef VT_RNW_T
#undef VT_RNW
    return vt_check_launch();
  }
  if (xdt == VT_F32 && D >= 512 && (rows >= 256 || D >= 1024) && (ldx % 4) == 0 && (ldy % 4) == 0) {    // block per row (also for few wide rows: latency)
    if (ydt == VT_F32) hipLaunchKernelGGL((rownorm_block_kernel<float, 2>), dim3(rows), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, w, b, D, eps, mode, range_flag);
    else if (ydt == VT_F16) hipLaunchKernelGGL((rownorm_block_kernel<half_t, 2>), dim3(rows), dim3(256), 0, s, (const float*)x, ldx, (half_t*)y, ldy, w, b, D, eps, mode, range_flag);
    else hipLaunchKernelGGL((rownorm_block_kernel<bf16_t, 2>), dim3(rows), dim3(256), 0, s, (const float*)x, ldx, (bf16_t*)y, ldy, w, b, D, eps, mode, range_flag);
    return vt_check_launch();
  }
  dim3 grid((rows + 3) / 4);
  DISPATCH_T(xdt, TI, DISPATCH_T(ydt, TO, {
    if (D <= 1024) hipLaunchKernelGGL((rownorm_kernel<TI, TO, 4>), grid, dim3(256), 0, s, (const TI*)x, ldx, (TO*)y, ldy, w, b, rows, D, eps, mode, range_flag);
    else hipLaunchKernelGGL((rownorm_kernel<TI, TO, 8>), grid, dim3(256), 0, s, (const TI*)x, ldx, (TO*)y, ldy, w, b, rows, D, eps, mode, range_flag);
  }))
  return vt_check_launch();
}

int vt_k_headnorm(void* x, int dt, long tok_stride, int heads, long tokens, const float* w, float eps, int mode, hipStream_t s) {
  const long rows = tokens * heads;
  if (rows <= 0) return VT_ERR_ARG;
  DISPATCH_T(dt, T, hipLaunchKernelGGL((headnorm_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (T*)x, tok_stride, heads, rows, w, eps, mode))
  return vt_check_launch();
}

int vt_k_slab_reduce(const float* slabs, int S, long slab_stride, int M, int N, const float* bias, int act, const float* colscale,
                     const void* residual, long ldr, void* out, int odt, long ldo, const float* hn_w0, const float* hn_w1, int hn_c0, int hn_c1,
                     float hn_eps, int hn_mode, hipStream_t s, const void* pf_ptr, size_t pf_bytes) {
  if (S < 1 || (N & 3) || M <= 0 || (hn_w0 && (N & 63))) return VT_ERR_ARG;
  const long n = (long)M * (N >> 2);
  if (pf_bytes >= (1ul << 31)) pf_ptr = nullptr;
  dim3 grid = g1(n);
  const int pf_block0 = (int)grid.x;
  if (pf_ptr) grid.x += VT_PF_BLOCKS;
  DISPATCH_T(odt, TO, hipLaunchKernelGGL((slab_reduce_kernel<TO>), grid, dim3(256), 0, s, slabs, S, slab_stride, M, N, bias, act, colscale,
                                         (const TO*)residual, ldr, (TO*)out, ldo, hn_w0, hn_w1, hn_c0, hn_c1, hn_eps, hn_mode, pf_ptr, (unsigned)pf_bytes, pf_block0))
  return vt_check_launch();
}

int vt_k_slab_reduce_norm(const float* slabs, int S, long slab_stride, int M, int N, const float* bias, float* x, long ldx, const float* w,
                          const float* b, float eps, int mode, void* xn, int xn_dt, long ldxn, hipStream_t s, const void* pf_ptr, size_t pf_bytes) {
  if (S < 1 || (N & 3) || N > 2048 || M <= 0 || (ldx & 3) || (ldxn & 3)) return VT_ERR_ARG;
  if (pf_bytes >= (1ul << 31)) pf_ptr = nullptr;
  const int grid = M + (pf_ptr ? VT_PF_BLOCKS : 0);
  DISPATCH_T(xn_dt, TN, hipLaunchKernelGGL((slab_reduce_norm_kernel<TN, 2>), dim3(grid), dim3(256), 0, s, slabs, S, slab_stride, N, bias, x, ldx, w, b, eps,
                                           mode, (TN*)xn, ldxn, pf_ptr, (unsigned)pf_bytes, M))
  return vt_check_launch();
}

int vt_k_groupnorm(const VtGnParams& p, hipStream_t s) {
  if (p.C % p.ngroups) return VT_ERR_ARG;
  const int n = (p.C / p.ngroups) * p.T;
  const size_t smem = (size_t)n * sizeof(float);
  if (smem > 64 * 1024) return VT_ERR_UNSUPPORTED;
  const int units = p.nets * p.B * p.ngroups;
  DISPATCH_T(p.out_dtype, TO, hipLaunchKernelGGL((gn_kernel<TO>), dim3(units), dim3(256), smem, s, p))
  return vt_check_launch();
}

int vt_k_sinusoid(const float* t, float t_host, void* out, int odt, int B, int dim, int nets, long net_stride, int rdt_style, hipStream_t s) {
  DISPATCH_T(odt, TO, hipLaunchKernelGGL((sinusoid_kernel<TO>), g1((long)B * dim), dim3(256), 0, s, t, t_host, (TO*)out, B, dim, nets, net_stride,
                                         rdt_style, rdt_style ? 0.f : 1.f))
  return vt_check_launch();
}

int vt_k_act_copy(const void* in, int idt, long ldi, void* out, int odt, long ldo, int rows, int cols, int act, hipStream_t s) {
  DISPATCH_T(idt, TI, DISPATCH_T(odt, TO, hipLaunchKernelGGL((act_copy_kernel<TI, TO>), g1((long)rows * cols), dim3(256), 0, s, (const TI*)in, ldi,
                                                              (TO*)out, ldo, rows, cols, act)))
  return vt_check_launch();
}

int vt_k_swiglu(void* h, int dt, long ld, long rows, int F, hipStream_t s, unsigned* range_flag) {
  const int V = (dt == VT_F32) ? 4 : 8;
  if (F % V || ld % V || rows <= 0) return VT_ERR_ARG;
  DISPATCH_T(dt, T, hipLaunchKernelGGL((swiglu_kernel<T>), g1(rows * (F / V)), dim3(256), 0, s, (T*)h, ld, rows, F, range_flag))
  return vt_check_launch();
}

int vt_k_sde_update(float* x, const float* v, const float* sc, const float* z, long n, float dt, float gi, float gdg, float eps,
                    float noise_scale, float d, float score_eps, int backward, hipStream_t s) {
  hipLaunchKernelGGL(sde_update_kernel, g1(n), dim3(256), 0, s, x, v, sc, z, n, dt, gi, gdg, eps, noise_scale, d, score_eps, backward);
  return vt_check_launch();
}

int vt_k_actnorm(const float* in, float* out, const float* mins, const float* maxs, long n, int dim, float pad, int denorm, hipStream_t s) {
  hipLaunchKernelGGL(actnorm_kernel, g1(n), dim3(256), 0, s, in, out, mins, maxs, n, dim, pad, denorm);
  return vt_check_launch();
}

int vt_k_pad_cols(const float* in, int cin, void* out, int odt, int cout, long rows, hipStream_t s) {
  DISPATCH_T(odt, TO, hipLaunchKernelGGL((pad_cols_kernel<TO>), g1(rows * cout), dim3(256), 0, s, in, cin, (TO*)out, cout, rows))
  return vt_check_launch();
}

int vt_k_place_cols(const void* src, int sdt, long lds_, void* out, int odt, long ldo, int off, int rows, int cols, hipStream_t s) {
  DISPATCH_T(sdt, TI, DISPATCH_T(odt, TO, hipLaunchKernelGGL((place_cols_kernel<TI, TO>), g1((long)rows * cols), dim3(256), 0, s, (const TI*)src, lds_,
                                                              (TO*)out, ldo, off, rows, cols)))
  return vt_check_launch();
}

int vt_k_bcast_row(const float* vec, float* out, long row_stride, int B, int D, hipStream_t s) {
  hipLaunchKernelGGL(bcast_row_kernel, g1((long)B * D), dim3(256), 0, s, vec, out, row_stride, B, D);
  return vt_check_launch();
}

int vt_k_imgstats(const void* img, int is_u8, long n, float pre_scale, int norm_mode, float* part, float* flags, hipStream_t s, float* flags_copy) {
  const int nb = 256;                  // the callers' partials buffers hold 256 x 2 floats per camera
  if (is_u8) hipLaunchKernelGGL((imgstat_partial_kernel<uint8_t>), dim3(nb), dim3(256), 0, s, (const uint8_t*)img, n, part);
  else hipLaunchKernelGGL((imgstat_partial_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)img, n, part);
  hipLaunchKernelGGL(imgstat_final_kernel, dim3(1), dim3(64), 0, s, part, nb, n, pre_scale, norm_mode, flags, flags_copy);
  return vt_check_launch();
}

int vt_k_patchify(const void* img, int is_u8, int nhwc, int B, int res, int grid_, int kpad, const float* flags, void* out, int odt, hipStream_t s) {
  if (kpad < 588) return VT_ERR_ARG;
  dim3 grid(B * grid_);
  DISPATCH_T(odt, TO, {
    if (is_u8) hipLaunchKernelGGL((patchify_kernel<uint8_t, TO>), grid, dim3(256), 0, s, (const uint8_t*)img, nhwc, B, res, grid_, kpad, flags, (TO*)out);
    else hipLaunchKernelGGL((patchify_kernel<float, TO>), grid, dim3(256), 0, s, (const float*)img, nhwc, B, res, grid_, kpad, flags, (TO*)out);
  })
  return vt_check_launch();
}

int vt_k_lstm_cell(const float* gi, const float* gh, float* h, float* c, int B, int H, hipStream_t s) {
  hipLaunchKernelGGL(lstm_cell_kernel, g1((long)B * H), dim3(256), 0, s, gi, gh, h, c, B, H);
  return vt_check_launch();
}

int vt_k_axpby3(float* x, const void* m0, const void* m1, int mdt, float a, float b0, float b1, long n, hipStream_t s) {
  DISPATCH_T(mdt, TM, hipLaunchKernelGGL((axpby3_kernel<TM>), g1(n), dim3(256), 0, s, x, (const TM*)m0, (const TM*)m1, a, b0, b1, n))
  return vt_check_launch();
}
