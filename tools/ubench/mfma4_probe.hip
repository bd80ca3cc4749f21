// Probe of v_mfma_f32_4x4x4_16b_bf16 (16 blocks of 4x4x4) operand / result layout on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma4_probe.hip -o tools/ubench/mfma4_probe && tools/ubench/mfma4_probe
// Hypothesis H1: block = lane / 4;  D[lane][r] = sum_k A[4*(lane/4) + r][k] * B[lane][k]   (rows from the A lanes of the block, column = own B lane)
// Hypothesis H2: D[lane][r] = sum_k A[lane][k] * B[4*(lane/4) + r][k]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
__global__ void probe(const uint16_t* A, const uint16_t* B, float* D) {
  const int lane = threadIdx.x;
  short4_t a, b;
  for (int k = 0; k < 4; ++k) { a[k] = (short)A[lane * 4 + k]; b[k] = (short)B[lane * 4 + k]; }
  float4_t c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[lane * 4 + r] = c[r];
}
int main() {
  uint16_t hA[256], hB[256];
  float fA[256], fB[256], hD[256];
  srand(3);
  for (int i = 0; i < 256; ++i) { fA[i] = (float)(rand() % 15 - 7); fB[i] = (float)(rand() % 15 - 7); hA[i] = f2bf(fA[i]); hB[i] = f2bf(fB[i]); }
  uint16_t *dA, *dB; float* dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 4; ++r) {
      float h1 = 0, h2 = 0;
      for (int k = 0; k < 4; ++k) {
        h1 += fA[(4 * (lane / 4) + r) * 4 + k] * fB[lane * 4 + k];
        h2 += fA[lane * 4 + k] * fB[(4 * (lane / 4) + r) * 4 + k];
      }
      bad1 += h1 != hD[lane * 4 + r];
      bad2 += h2 != hD[lane * 4 + r];
    }
  printf("H1 mismatches %d, H2 mismatches %d (of 256)\n", bad1, bad2);
  return 0;
}
