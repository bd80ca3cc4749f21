// pk_fp32_probe.hip — does the auto-vectorised (packed fp32) form of  y = (x - mean) * rstd * gamma + beta ; mish ; y = s * y + o  on float4
// values read from LDS give the same bits when it is evaluated twice from the same LDS inputs while two workgroups share a CU?  (The prologue of csrc/vt_uconv.hip did not,
// "packed fp32 under co-residency".)
//   hipcc --offload-arch=gfx950 -O3 -o pk_probe tools/ubench/pk_fp32_probe.hip && ./pk_probe [blocks_per_cu]
// Every thread evaluates the statements twice (an LDS write and a barrier in between) and compares bit patterns; mismatches are counted.
// Build a second time with  -Xclang -target-feature -Xclang -packed-fp32-ops  for the scalar instruction selection.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(2))) float float2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ float mish(float x) {
  if (x > 20.0f) return x;
  const float n = __builtin_amdgcn_exp2f(x * 1.4426950408889634f), w = n * (n + 2.0f);
  return x * w * __builtin_amdgcn_rcpf(w + 2.0f);
}

// rows x cs floats of "stage", stats per row group, params; planes out.  Mirrors the P3 loop of uconv_kernel.
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, int* __restrict__ bad, int rows, int cs, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int c4n = cs >> 2, total4 = rows * c4n;
  float4_t* stage = reinterpret_cast<float4_t*>(smem);
  float2* stats = reinterpret_cast<float2*>(smem + (size_t)total4 * 16);
  float4_t* par4 = reinterpret_cast<float4_t*>(smem + (size_t)total4 * 16 + 2048);
  char* plane = smem + (size_t)total4 * 16 + 2048 + (size_t)4 * c4n * 16;
  const int pitch = cs * 2 + 16;
  int nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const float* s = src + ((size_t)(blockIdx.x * 7 + it) % 64) * 4096;
    for (int e = tid; e < total4; e += 256) stage[e] = *reinterpret_cast<const float4_t*>(s + (e * 4) % 4096);
    if (tid < 64) stats[tid] = make_float2(s[tid] * 0.1f, 1.0f + 0.01f * s[tid + 64]);
    for (int e = tid; e < 4 * c4n; e += 256) par4[e] = *reinterpret_cast<const float4_t*>(s + ((e * 4 + 512) % 4096));
    __syncthreads();
    for (int e = tid; e < total4; e += 256) {
      const int r = e / c4n, c4 = e - r * c4n;
      const float2 st = stats[r & 63];
      float4_t y = stage[e];
      // vector form (hipcc packs it: v_pk_mul_f32 / v_pk_fma_f32)
      float4_t yv = (y - st.x) * st.y * par4[c4] + par4[c4n + c4];
#pragma unroll
      for (int i = 0; i < 4; ++i) yv[i] = mish(yv[i]);
      yv = par4[2 * c4n + c4] * yv + par4[3 * c4n + c4];
      const uint32_t h0 = pk_bf16(yv[0], yv[1]), h1 = pk_bf16(yv[2], yv[3]);
      *reinterpret_cast<uint2*>(plane + r * pitch + c4 * 8) = make_uint2(h0, h1);
    }
    __syncthreads();
    for (int e = tid; e < total4; e += 256) {
      const int r = e / c4n, c4 = e - r * c4n;
      const float2 st = stats[r & 63];
      float4_t y = stage[e];
      // the SAME statements again (LDS was written and a barrier passed in between: nothing can be reused from the first pass)
      float4_t yv = (y - st.x) * st.y * par4[c4] + par4[c4n + c4];
#pragma unroll
      for (int i = 0; i < 4; ++i) yv[i] = mish(yv[i]);
      yv = par4[2 * c4n + c4] * yv + par4[3 * c4n + c4];
      const uint32_t h0 = pk_bf16(yv[0], yv[1]), h1 = pk_bf16(yv[2], yv[3]);
      const uint2 w = *reinterpret_cast<const uint2*>(plane + r * pitch + c4 * 8);
      nbad += (w.x != h0) + (w.y != h1);
    }
    __syncthreads();
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 2;
  const int rows = 32, cs = 64;
  const size_t lds = (size_t)rows * cs * 4 + 2048 + (size_t)4 * (cs / 4) * 16 + (size_t)rows * (cs * 2 + 16);
  float* src; int* bad;
  hipMalloc(&src, 64 * 4096 * 4); hipMalloc(&bad, 4);
  float* h = (float*)malloc(64 * 4096 * 4);
  srand(1);
  for (int i = 0; i < 64 * 4096; ++i) h[i] = (float)rand() / RAND_MAX * 4.0f - 2.0f;
  hipMemcpy(src, h, 64 * 4096 * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 4);
  for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL(probe, dim3(256 * per_cu), dim3(256), per_cu == 1 ? 100 * 1024 : lds, 0, src, bad, rows, cs, 8);
  int nb = 0;
  hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
  printf("blocks per CU %d: %d mismatching words (second evaluation vs first) over 50 launches x %d blocks x 8 iterations\n", per_cu, nb, 256 * per_cu);
  return 0;
}
