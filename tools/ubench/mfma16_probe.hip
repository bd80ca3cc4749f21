// layout probe of v_mfma_f32_16x16x16_{bf16,f16}: A = "identity" (A[i][k] = k == i), B[j][k] = j*16 + k (asymmetric) -> expect D[i][j] = B[j][i]
// under the assumed mapping: lane l holds A[l&15][(l>>4)*4 .. +3], B[l&15][(l>>4)*4 .. +3]; D[(l>>4)*4 + r][l&15].
//   hipcc --offload-arch=gfx950 -O2 mfma16_probe.hip -o mfma16_probe && ./mfma16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
__device__ short f2bf(float f) { return (short)(__float_as_uint(f) >> 16); }
__global__ void probe(float* out) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  short4_t a, b; half4_t ah, bh;
  for (int j = 0; j < 4; ++j) {
    const int k = g * 4 + j;
    const float av = k == r ? 1.f : 0.f, bv = (float)(r * 16 + k);
    a[j] = f2bf(av); b[j] = f2bf(bv); ah[j] = (_Float16)av; bh[j] = (_Float16)bv;
  }
  float4_t c = {0, 0, 0, 0}, d = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, d, 0, 0, 0);
  for (int j = 0; j < 4; ++j) { out[l * 4 + j] = c[j]; out[256 + l * 4 + j] = d[j]; }
}
int main() {
  float* dv; float h[512];
  (void)hipMalloc(&dv, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dv);
  (void)hipMemcpy(h, dv, sizeof(h), hipMemcpyDeviceToHost);
  int bad[2] = {0, 0};
  for (int t = 0; t < 2; ++t)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int i = (l >> 4) * 4 + j, jj = l & 15;            // D[i][jj] expected = B[jj][i] = jj*16 + i
        if (h[t * 256 + l * 4 + j] != (float)(jj * 16 + i)) { if (bad[t] < 4) printf("type %d lane %d r %d: got %g want %d\n", t, l, j, h[t * 256 + l * 4 + j], jj * 16 + i); ++bad[t]; }
      }
  printf("mfma 16x16x16 layout: bf16 mismatches %d, f16 mismatches %d (0 = the assumed mapping holds)\n", bad[0], bad[1]);
  return 0;
}
