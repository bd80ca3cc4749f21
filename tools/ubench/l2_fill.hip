// Micro-benchmark: per-CU rate of pulling L2-resident data, (a) by LDS-DMA (global_load_lds_dwordx4), (b) by global_load_dwordx4
// into registers (+ ds_write_b128), (c) half and half.  One 256-thread block per CU, each cycling over its own 96 KiB region.
//   hipcc --offload-arch=gfx950 -O3 l2_fill.hip -o l2_fill && ./l2_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int REGION = 96 * 1024, ITERS = 400;

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (size_t)blockIdx.x * REGION;
  float acc = 0.f;
  for (int it = 0; it < ITERS; ++it) {
    // 32 KiB per iteration: 32 pieces of 1 KiB, 8 per wave
    const int off0 = (it % 3) * 32768;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int piece = wave * 8 + e;
      const char* p = base + off0 + piece * 1024 + lane * 16;
      const bool dma = MODE == 0 || (MODE == 2 && (e & 1));
      if (dma) __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(smem + piece * 1024), 16, 0, 0);
      else { const uint4 v = *reinterpret_cast<const uint4*>(p); *reinterpret_cast<uint4*>(smem + piece * 1024 + lane * 16) = v; }
    }
    __syncthreads();
    acc += *reinterpret_cast<const float*>(smem + tid * 4);
    __syncthreads();
  }
  if (acc == 1234.5f) out[0] = acc;
}

int main() {
  const int blocks = 256;
  char* src; float* out;
  hipMalloc(&src, (size_t)blocks * REGION); hipMalloc(&out, 4);
  hipMemset(src, 1, (size_t)blocks * REGION);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, out);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, out);
      else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, src, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)blocks * ITERS * 32768;
      if (rep) printf("mode %d (%s): %.3f ms  %.2f TB/s  %.1f B/clk/CU @2.1GHz\n", mode, mode == 0 ? "LDS-DMA" : mode == 1 ? "global_load + ds_write" : "half/half",
                      ms, bytes / ms / 1e9, bytes / blocks / (ms * 1e-3) / 2.1e9);
    }
  }
  return 0;
}
