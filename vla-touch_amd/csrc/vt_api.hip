// vt_api.hip — C-ABI entry points for the primitives (unit-test hooks) and the MFMA layout self test.
#include <string.h>
#include "vt_common.h"
#include "vt_kernels.h"
#include "vt_host.h"
#include "../../include/vlatouch.h"

extern "C" {

const char* vt_last_error(void) { return vt_errbuf(); }
int vt_version(void) { return 100; }

int vt_gemm(const void* params, vt_stream_t stream) {
  if (!params) return vt_fail(VT_ERR_ARG, "vt_gemm: null params");
  const int r = vt_gemm_launch(*reinterpret_cast<const VtGemmParams*>(params), (hipStream_t)stream);
  if (r) vt_fail(r, "vt_gemm: rejected (alignment: K, lda, ldw multiples of 8 bf16 / 4 fp32; conv: cin likewise) or launch failure");
  return r;
}
int vt_attention(const void* params, vt_stream_t stream) {
  if (!params) return vt_fail(VT_ERR_ARG, "vt_attention: null params");
  const int r = vt_attn_launch(*reinterpret_cast<const VtAttnParams*>(params), (hipStream_t)stream);
  if (r) vt_fail(r, "vt_attention: bad strides/sizes or launch failure");
  return r;
}
int vt_groupnorm(const void* params, vt_stream_t stream) {
  if (!params) return vt_fail(VT_ERR_ARG, "vt_groupnorm: null params");
  const int r = vt_k_groupnorm(*reinterpret_cast<const VtGnParams*>(params), (hipStream_t)stream);
  if (r) vt_fail(r, "vt_groupnorm: unsupported group size or launch failure");
  return r;
}
int vt_rownorm(const void* x, int xdt, long ldx, void* y, int ydt, long ldy, const float* w, const float* b, int rows, int D, float eps,
               int mode, vt_stream_t stream) {
  const int r = vt_k_rownorm(x, xdt, ldx, y, ydt, ldy, w, b, rows, D, eps, mode, (hipStream_t)stream);
  if (r) vt_fail(r, "vt_rownorm: D must be a multiple of 4 and <= 2048");
  return r;
}
int vt_headnorm(void* x, int dt, long tok_stride, int heads, long tokens, const float* w, float eps, int mode, vt_stream_t stream) {
  return vt_k_headnorm(x, dt, tok_stride, heads, tokens, w, eps, mode, (hipStream_t)stream);
}
int vt_action_normalize(const float* in, float* out, const float* mins, const float* maxs, long n, int dim, float padding_factor,
                        int denormalize, vt_stream_t stream) {
  if (!in || !out || !mins || !maxs || n < 0 || dim < 1) return vt_fail(VT_ERR_ARG, "vt_action_normalize: bad argument");
  if (n == 0) return VT_OK;
  return vt_k_actnorm(in, out, mins, maxs, n, dim, padding_factor, denormalize, (hipStream_t)stream);
}

}  // extern "C"

// ---------------------------------------------------------------- MFMA layout self test
namespace {
template <typename T>
__global__ void mfma_selftest_kernel(float* err) {
  // D = Aop * Bop with Aop = W-role tile "I" (16x32 with ones on the diagonal of the first 16 k) and an
  // ASYMMETRIC second operand X[col][k] = col*100 + k: expect D[row=i][col=j] = X[j][i].
  const int lane = threadIdx.x, g = lane >> 4, l15 = lane & 15;
  Frag<T> a, b;
  float av[8], bv[8];
  for (int j = 0; j < 8; ++j) {
    const int k = g * 8 + j;
    av[j] = (k == l15) ? 1.f : 0.f;
    bv[j] = (float)(l15 * 100 + k);
  }
  if constexpr (sizeof(T) == 2) {
    for (int j = 0; j < 8; ++j) { a.v[j] = (short)f2bf(av[j]); b.v[j] = (short)f2bf(bv[j]); }
  } else {
    for (int j = 0; j < 8; ++j) { a.v[j] = av[j]; b.v[j] = bv[j]; }
  }
  float4_t acc = {0.f, 0.f, 0.f, 0.f};
  mma16(acc, a, b);
  float e = 0.f;
  for (int r = 0; r < 4; ++r) {
    const int row = g * 4 + r, col = l15;
    float expect = (float)(col * 100 + row);
    if constexpr (sizeof(T) == 2) expect = bf2f(f2bf(expect));
    e = fmaxf(e, fabsf(acc[r] - expect));
  }
  e = wave_max(e);
  if (lane == 0) *err = e;
}
}  // namespace

extern "C" int vt_selftest_mfma(float* out_err, vt_stream_t stream) {
  if (!out_err) return vt_fail(VT_ERR_ARG, "vt_selftest_mfma: null");
  hipLaunchKernelGGL((mfma_selftest_kernel<bf16_t>), dim3(1), dim3(64), 0, (hipStream_t)stream, out_err);
  hipLaunchKernelGGL((mfma_selftest_kernel<float>), dim3(1), dim3(64), 0, (hipStream_t)stream, out_err + 1);
  return vt_check_launch();
}

// ---------------------------------------------------------------- Gaussian draws on the device (replaces the reference's torch.randn /
// torch.randn_like draws, rdt_runner.py:136 and bridge_model.py:372, when the caller does not inject its own noise): counter-based
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller, four normals per counter.  The (key, counter) pair lives in DEVICE memory
// (`state` = 2 x uint64: key, next counter) and is advanced by a one-thread kernel behind the draw, so a captured hipGraph yields fresh
// noise on every replay without a host round trip.
namespace {
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t k0, const uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}
// out[i] = N(0, 1) (optionally rounded to the bf16 grid: `noisy_action` lives in bf16 in the reference's bf16 mode), 4 values per thread
__global__ void randn_kernel(float* __restrict__ out, const long n, const unsigned long long* __restrict__ state, const int round_bf16) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q * 4 >= n) return;
  const unsigned long long key = state[0], ctr = state[1] + (unsigned long long)q;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
  float v[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0, 1): log never sees 0
    const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    v[2 * h] = r * cs; v[2 * h + 1] = r * sn;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (round_bf16) v[j] = bf2f(f2bf(v[j]));
    if (q * 4 + j < n) out[q * 4 + j] = v[j];
  }
}
__global__ void randn_advance_kernel(unsigned long long* state, const unsigned long long by) { state[1] += by; }
// out[b][t][d] (fp32) = in[b][t][d] for t < T, d < Dd of in [B][Tin][Din] (adt): the slice + cast between the RDT chunk and the controller
template <typename TI>
__global__ void slice_cast_kernel(const TI* __restrict__ in, float* __restrict__ out, int B, int Tin, int Din, int T, int Dd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * T * Dd) return;
  const int d = (int)(i % Dd);
  const long r = i / Dd;
  const int t = (int)(r % T), b = (int)(r / T);
  out[i] = Elem<TI>::to_f(in[((long)b * Tin + t) * Din + d]);
}
}  // namespace

extern "C" int vt_randn(float* out, long n, void* state, int round_bf16, vt_stream_t stream) {
  if (!out || !state || n < 0) return vt_fail(VT_ERR_ARG, "vt_randn: null argument");
  if (n == 0) return VT_OK;
  const long quads = (n + 3) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, (const unsigned long long*)state, round_bf16);
  hipLaunchKernelGGL(randn_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)state, (unsigned long long)quads);
  return vt_check_launch();
}
extern "C" int vt_slice_cast(const void* in, int idt, float* out, int B, int Tin, int Din, int T, int Dd, vt_stream_t stream) {
  if (!in || !out || B < 1 || T < 1 || Dd < 1 || T > Tin || Dd > Din) return vt_fail(VT_ERR_ARG, "vt_slice_cast: bad shape");
  const long n = (long)B * T * Dd;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (idt == VT_BF16) hipLaunchKernelGGL((slice_cast_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out, B, Tin, Din, T, Dd);
  else if (idt == VT_F32) hipLaunchKernelGGL((slice_cast_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)in, out, B, Tin, Din, T, Dd);
  else return vt_fail(VT_ERR_UNSUPPORTED, "vt_slice_cast: bf16 or fp32 input");
  return vt_check_launch();
}

// out[rows][cols] (odt, row stride ldo) = in[rows][cols] (idt, row stride ldi): dtype conversion between pipeline stages (e.g. the fp32 image
// tokens of the SigLIP tower -> the bf16 `img_tokens` of RDTRunner.predict_action; franka_model_eef.py:286-288 does `.to(dtype)`)
extern "C" int vt_cast(const void* in, int idt, long ldi, void* out, int odt, long ldo, int rows, int cols, vt_stream_t stream) {
  if (!in || !out || rows < 1 || cols < 1) return vt_fail(VT_ERR_ARG, "vt_cast: bad argument");
  const int r = vt_k_act_copy(in, idt, ldi, out, odt, ldo, rows, cols, VT_ACT_NONE, (hipStream_t)stream);
  if (r) vt_fail(r, "vt_cast: unsupported dtype pair");
  return r;
}
