#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short short4_t;
__global__ void probe(int* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;      // element index = row * 64 + col
  __syncthreads();
  // each lane supplies the address of 4 contiguous b16: row = lane / 4 (0..15), cols (lane % 4) * 4 .. +3
  const int row = lane >> 2, c4 = (lane & 3) * 4;
  __attribute__((address_space(3))) short4_t* p = (__attribute__((address_space(3))) short4_t*)(lds + row * stride_elems + c4);
  short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (int)(uint16_t)v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 64);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane address: row = lane/4, cols (lane%%4)*4..+3 of a [16][64] b16 tile; value = row*64+col\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (r%2d,c%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
    printf("\n");
  }
  return 0;
}
