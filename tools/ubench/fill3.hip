// Micro-benchmark: does the LDS-DMA fill share the LDS port badly with fragment reads?  The access pattern of the per-denoise-step RDT
// Linear (fill2.hip: 224 blocks, one per CU, 160-row A panel + 128-row W panel per k-tile of 64 bf16 = 36 KiB), now WITH the fragment
// reads of gemm_ppk_kernel (4 waves x 18 ds_read_b128 = 72 KiB per k-tile, alternating wave groups) but still without MFMAs:
//   mode 0: fill by LDS-DMA (global_load_lds), no reads                          (= fill2 mode 0)
//   mode 1: fill by LDS-DMA + the fragment reads of the k-tile that landed two tiles ago
//   mode 2: fill by full-line register loads + ds_write_b128 one tile later, no reads
//   mode 3: register loads + ds_write_b128 + the same fragment reads
//   hipcc --offload-arch=gfx950 -O3 fill3.hip -o fill3 && ./fill3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int K = 2048, PITCH = K * 2, NKT = 32, REPS = 8;
constexpr int AROWS = 160, WROWS = 128;
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ A, const char* __restrict__ W, unsigned* __restrict__ out, long long* __restrict__ cyc) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 36864];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tm = blockIdx.x / 16, tn = blockIdx.x % 16;
  const char* Ab = A + (size_t)tm * AROWS * PITCH;
  const char* Wb = W + (size_t)tn * WROWS * PITCH;
  constexpr bool DMA = MODE < 2, READS = (MODE & 1) != 0;
  uint4 acc = make_uint4(0, 0, 0, 0);
  // piece q of a k-tile (36 x 1 KiB): wave w owns q = w, w+8, w+16, w+24 (+ w+32 for w < 4)
  auto src = [&](int q, int kt) {
    const char* base = q < 20 ? Ab : Wb;
    const int row8 = q < 20 ? q : q - 20;
    return base + (size_t)(row8 * 8 + (lane >> 3)) * PITCH + kt * 128 + (lane & 7) * 16;
  };
  auto frag_reads = [&](int stage, int grp) {          // the group's 4 waves read 18 fragments each (rows spread over the tile)
    if ((wave >> 2) != grp) return;
    const char* s = smem + stage * 36864;
#pragma unroll
    for (int f = 0; f < 18; ++f) {
      const int row = ((wave & 3) * 18 + f) * 4 % 288 + (lane & 15) % 4;     // any in-range row; 16-B chunk by lane group
      const uint4 v = *reinterpret_cast<const uint4*>(s + row * 128 + (((lane >> 4) + f) & 7) * 16);
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  };
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < REPS; ++rep) {
    if constexpr (DMA) {
      for (int kt = 0; kt < NKT; ++kt) {
        const int st = kt & 3;
#pragma unroll
        for (int e = 0; e < 5; ++e) {
          const int q = wave + 8 * e;
          if (q < 36) __builtin_amdgcn_global_load_lds((glb_void*)src(q, kt), (lds_void*)(smem + st * 36864 + q * 1024), 16, 0, 0);
        }
        WAIT_VM(10);                                   // ~2 k-tiles stay in flight
        __builtin_amdgcn_s_barrier();
        if (READS && kt >= 2) frag_reads((kt - 2) & 3, kt & 1);
      }
      WAIT_VM(0);
    } else {
      uint4 r[2][5];                                   // two k-tiles in registers (loaded one and two iterations ago)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 5; ++e) r[s][e] = make_uint4(0, 0, 0, 0);
      for (int kt0 = 0; kt0 < NKT; kt0 += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int kt = kt0 + s, st = kt & 3;
          // write the tile loaded two iterations ago (its loads have had two iterations to land), then reload the slot
#pragma unroll
          for (int e = 0; e < 5; ++e) {
            const int q = wave + 8 * e;
            if (q < 36) {
              if (kt >= 2) *reinterpret_cast<uint4*>(smem + ((kt - 2) & 3) * 36864 + q * 1024 + lane * 16) = r[s][e];
              r[s][e] = *reinterpret_cast<const uint4*>(src(q, kt));
            }
          }
          __builtin_amdgcn_s_barrier();
          if (READS && kt >= 3) frag_reads((kt - 3) & 3, kt & 1);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 5; ++e) { acc.x ^= r[s][e].x; acc.y ^= r[s][e].y; acc.z ^= r[s][e].z; acc.w ^= r[s][e].w; }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc.x + smem[tid];
}

int main() {
  const int blocks = 224;
  const size_t abytes = (size_t)14 * AROWS * PITCH, wbytes = (size_t)16 * WROWS * PITCH;
  char *A, *W; unsigned* out; long long* cyc;
  hipMalloc(&A, abytes); hipMalloc(&W, wbytes); hipMalloc(&out, 4); hipMalloc(&cyc, blocks * 8);
  hipMemset(A, 1, abytes); hipMemset(W, 2, wbytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"LDS-DMA fill", "LDS-DMA fill + fragment reads", "register fill + ds_write", "register fill + ds_write + fragment reads"};
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        case 1: hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        case 2: hipLaunchKernelGGL(fill<2>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        default: hipLaunchKernelGGL(fill<3>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("mode %d %-44s: %.1f us per 32 k-tiles (%.0f ns per k-tile)\n", mode, names[mode], best * 1e3 / REPS, best * 1e6 / REPS / NKT);
  }
  return 0;
}
