// Micro-benchmark: sustained per-CU operand fill rate in the access pattern of the per-denoise-step RDT Linears
// (M = 2144, N = K = 2048 bf16: 224 blocks, one per CU, each walking its own 160-row A panel and 128-row W panel along k;
// both matrices (17 MB) stay L2 / Infinity-Cache resident), PIPELINED (counted vmcnt, 3 k-tiles in flight), no MFMA work:
//   mode 0: everything by LDS-DMA (buffer-less global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave instruction)
//   mode 1: everything by global_load_dwordx4 into registers, full 128-B lines (8 rows x 128 B per instruction)
//   mode 2: the same bytes by fragment-shaped register loads (16 rows x 64 B per instruction = the MFMA operand layout)
//   mode 3: A panel by LDS-DMA + W panel by fragment-shaped register loads (each W row block loaded by two waves)
//   mode 4: A panel by LDS-DMA + W panel by full-line register loads (each W row block loaded by two waves)
//   hipcc --offload-arch=gfx950 -O3 fill2.hip -o fill2 && ./fill2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int K = 2048, PITCH = K * 2, NKT = 32, REPS = 8;
constexpr int AROWS = 160, WROWS = 128;

#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ A, const char* __restrict__ W, unsigned* __restrict__ out, long long* __restrict__ cyc) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 36 * 1024];      // 4 stages x 36 KiB (A 20 KiB | W 16 KiB) = 144 KiB: one block per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tm = blockIdx.x / 16, tn = blockIdx.x % 16;
  const char* Ab = A + (size_t)tm * AROWS * PITCH;
  const char* Wb = W + (size_t)tn * WROWS * PITCH;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const long long t0 = __builtin_amdgcn_s_memtime();
  // pieces of a k-tile: 36 x (8 rows x 128 B): 20 of A, 16 of W.  Waves 0..3 take 5 pieces each in modes that split by 4, else 8 waves x 4.5
  auto dma_piece = [&](const char* base, int row8, int kt, int stage, int slot) {
    const char* p = base + (size_t)(row8 * 8 + (lane >> 3)) * PITCH + kt * 128 + (lane & 7) * 16;
    __builtin_amdgcn_global_load_lds((glb_void*)p, (lds_void*)(smem + stage * 36864 + slot * 1024), 16, 0, 0);
  };
  auto line_load = [&](const char* base, int row8, int kt) {
    return *reinterpret_cast<const uint4*>(base + (size_t)(row8 * 8 + (lane >> 3)) * PITCH + kt * 128 + (lane & 7) * 16);
  };
  auto frag_load = [&](const char* base, int row16, int kt, int ks) {      // 16 rows x 64 B
    return *reinterpret_cast<const uint4*>(base + (size_t)(row16 * 16 + (lane & 15)) * PITCH + kt * 128 + ks * 64 + (lane >> 4) * 16);
  };
  for (int rep = 0; rep < REPS; ++rep) {
    if constexpr (MODE == 0) {
      // 36 pieces per k-tile: wave w issues pieces w, w+8, w+16, w+24 (+ w+32 for w < 4); keep ~3 k-tiles in flight
      for (int kt = 0; kt < NKT; ++kt) {
        const int st = kt & 3;
#pragma unroll
        for (int e = 0; e < 5; ++e) {
          const int q = wave + 8 * e;
          if (q < 20) dma_piece(Ab, q, kt, st, q);
          else if (q < 36) dma_piece(Wb, q - 20, kt, st, q);
        }
        WAIT_VM(10);
        __builtin_amdgcn_s_barrier();
      }
      WAIT_VM(0);
    } else if constexpr (MODE == 1 || MODE == 2) {
      uint4 r[3][5];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 5; ++e) r[s][e] = make_uint4(0, 0, 0, 0);
      for (int kt0 = 0; kt0 < NKT; kt0 += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int kt = kt0 + s;
          if (kt < NKT) {
#pragma unroll
            for (int e = 0; e < 5; ++e) {
              acc.x ^= r[s][e].x; acc.y ^= r[s][e].y; acc.z ^= r[s][e].z; acc.w ^= r[s][e].w;      // consume what this slot held (3 k-tiles old)
              const int q = wave + 8 * e;
              if (MODE == 1) {
                if (q < 20) r[s][e] = line_load(Ab, q, kt); else if (q < 36) r[s][e] = line_load(Wb, q - 20, kt);
              } else {
                // fragment-shaped: piece q -> (row16 = q / 2, ks = q % 2): the same 1 KiB per instruction
                if (q < 20) r[s][e] = frag_load(Ab, q >> 1, kt, q & 1); else if (q < 36) r[s][e] = frag_load(Wb, (q - 20) >> 1, kt, q & 1);
              }
            }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < 5; ++e) { acc.x ^= r[s][e].x; acc.y ^= r[s][e].y; acc.z ^= r[s][e].z; acc.w ^= r[s][e].w; }
    } else {
      // A by DMA: 20 pieces per k-tile, waves 0..7 issue 2.5 -> wave w: pieces w, w+8, (w+16 if < 20).  W by register loads: wave w loads the
      // 64-row half (w & 1) of the W tile = 8 KiB = 8 instructions (4 waves load each half: here 2x the duplication of the real kernel's
      // two, to keep every wave busy on every k-tile; the real kernel alternates k-tiles between its two wave groups)
      uint4 r[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[s][e] = make_uint4(0, 0, 0, 0);
      for (int kt0 = 0; kt0 < NKT; kt0 += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int kt = kt0 + s, st = kt & 3;
#pragma unroll
          for (int e = 0; e < 3; ++e) {
            const int q = wave + 8 * e;
            if (q < 20) dma_piece(Ab, q, kt, st, q);
          }
          // this wave's share of the W half: of its 8 instructions the wave pair (w>>1)&1 splits even/odd -> 4 each, so that per k-tile
          // the block issues 8 waves x 4 = 32 KiB of W register loads (each W line fetched twice, as in the real kernel)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc.x ^= r[s][e].x; acc.y ^= r[s][e].y; acc.z ^= r[s][e].z; acc.w ^= r[s][e].w;
            const int half = wave & 1, sub = ((wave >> 1) & 1) * 4 + e;      // sub 0..7 within the 64-row half
            if (MODE == 3) r[s][e] = frag_load(Wb, half * 4 + (sub >> 1), kt, sub & 1);
            else r[s][e] = line_load(Wb, half * 8 + sub, kt);
          }
        }
        __builtin_amdgcn_s_barrier();
      }
      WAIT_VM(0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc.x ^= r[s][e].x; acc.y ^= r[s][e].y; acc.z ^= r[s][e].z; acc.w ^= r[s][e].w; }
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
  const char* names[5] = {"all LDS-DMA", "all register, full lines", "all register, fragment-shaped", "A DMA + W fragment regs (W x2)", "A DMA + W full-line regs (W x2)"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f; long long cbest = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(fill<0>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        case 1: hipLaunchKernelGGL(fill<1>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        case 2: hipLaunchKernelGGL(fill<2>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        case 3: hipLaunchKernelGGL(fill<3>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
        default: hipLaunchKernelGGL(fill<4>, dim3(blocks), dim3(512), 0, 0, A, W, out, cyc); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[224]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      long long mx = 0; for (int i = 0; i < blocks; ++i) mx = h[i] > mx ? h[i] : mx;
      if (ms < best) { best = ms; cbest = mx; }
    }
    const double unique = (double)REPS * NKT * 36864;                       // bytes a CU needs per launch (A 20 KiB + W 16 KiB per k-tile)
    const double moved = mode >= 3 ? (double)REPS * NKT * (20480 + 32768) : unique;
    printf("mode %d %-34s: %.1f us per 32 k-tiles  | needed bytes %.1f B/memtime-tick/CU, moved %.1f | %.1f GB/s/CU needed  (ticks %lld, %.3f ms)\n", mode, names[mode],
           best * 1e3 / REPS, unique / cbest, moved / cbest, unique / (best * 1e-3) / 1e9, cbest, best);
  }
  return 0;
}
