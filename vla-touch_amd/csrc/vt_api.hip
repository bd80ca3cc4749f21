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
