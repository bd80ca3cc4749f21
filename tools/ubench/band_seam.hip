// Micro-benchmark (round 6): the BAND-LOCAL SEAM of a persistent RDT denoise kernel, in isolation.
//
// Every op of an RDT block is per-sample (attention mixes only the 67 rows of one episode), so a persistent kernel over the per-denoise-step Linears needs no grid-wide
// barrier: with the weights-in-registers tiling of csrc/vt_gemm_pw.hip (M = 2 144 rows = 14 bands of 160 rows, N = 2 048 = 16 tiles of 128 columns) the 16 workgroups
// of a band produce the band's 160 x 2 048 fp16 output (40 KiB each) and the SAME 16 workgroups consume all of it as the A operand of the next Linear.  A seam is then:
//   [publish my 160 x 128 tile] -> [arrive on the band's flags] -> [wait for the producers] -> [read the band: first k-tile = 160 x 64 fp16 = 20 KiB, all = 640 KiB]
// This program times exactly that, S seams back to back inside ONE launch of 224 workgroups (one per CU), with in-kernel s_memrealtime stamps (100 MHz) per phase:
//   mode 0  no handshake at all (reads whatever is there): the traffic-only baseline
//   mode 1  plain stores -> __threadfence() (release: buffer_wbl2) -> relaxed atomic flag ; consumer: relaxed sc1 poll + s_sleep -> acquire fence -> plain loads
//   mode 2  `sc1` (agent-scope write-through) stores -> asm s_waitcnt vmcnt(0) -> flag ; consumer: poll -> acquire fence -> plain loads   (MI355X_MICROARCH "publish-large")
//   mode 3  the same stores ; consumer: poll -> `sc0 sc1` loads, no acquire       mode 4: consumer `sc1` loads, no acquire
//   mode 5 / 6 / 7 = 2 / 3 / 4 with `sc0 sc1` (system-scope) stores
//   mode 8  SAME-XCD ONLY: plain stores (the lines stay dirty in the XCD's L2, which all 32 CUs of the XCD share) -> asm s_waitcnt vmcnt(0) -> flag ; consumer: poll ->
//           `sc1` loads (bypass this CU's L1, served by the shared L2).  No write-back, no invalidate: valid only when producer and consumer sit on one XCD, so with the
//           bands spread over the XCDs it MUST read stale words (the run shows how many) — a kernel using it has to derive band membership from HW_REG_XCC_ID
// wait form  A  one counter per band (all 16 arrived before anything is read)        F  one flag per producer: member j's slab is read when flag j is up, in order j = 0 .. 15
// placement  X  a band's 16 workgroups on ONE XCD (blockIdx % 8 = band % 8)           S  16 consecutive block ids = spread over all 8 XCDs
// Every word read is checked against the value the seam's producer must have written (mode 1 / 2): `bad` must be 0.
//   hipcc --offload-arch=gfx950 -O3 band_seam.hip -o band_seam && ./band_seam
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

constexpr int BANDS = 14, MEMBERS = 16, ROWS = 160, COLS = 2048, TCOLS = 128;
constexpr int SEAMS = 200;
constexpr int BAND_ELEMS = ROWS * COLS;                 // fp16 elements of a band (640 KiB)

__device__ __forceinline__ long long now() { return __builtin_amdgcn_s_memrealtime(); }
template <int BITS> __device__ __forceinline__ void st_wt(void* p, uint4 v) {      // BITS 2: `sc1` (agent scope), 3: `sc0 sc1` (system scope)
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  const u4 t = {v.x, v.y, v.z, v.w};
  // (s_nop: the hardware reads a > 8-byte store's data registers a few cycles after issue and the compiler, which inserts that wait state behind its own stores,
  //  does not look inside inline asm — without it the next iteration's VALU writes corrupted 20 % of the published words)
  if constexpr (BITS == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 2" ::"v"(p), "v"(t) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" ::"v"(p), "v"(t) : "memory");
}
// ten 16-byte loads with the given cache-coherence bits in flight together, retired by ONE wait inside the same asm statement (the compiler never sees a
// register whose load has not landed)
typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
#define LD10(BITS) \
  asm volatile("global_load_dwordx4 %0, %10, off " BITS "\n\tglobal_load_dwordx4 %1, %11, off " BITS "\n\tglobal_load_dwordx4 %2, %12, off " BITS "\n\t" \
               "global_load_dwordx4 %3, %13, off " BITS "\n\tglobal_load_dwordx4 %4, %14, off " BITS "\n\tglobal_load_dwordx4 %5, %15, off " BITS "\n\t" \
               "global_load_dwordx4 %6, %16, off " BITS "\n\tglobal_load_dwordx4 %7, %17, off " BITS "\n\tglobal_load_dwordx4 %8, %18, off " BITS "\n\t" \
               "global_load_dwordx4 %9, %19, off " BITS "\n\ts_waitcnt vmcnt(0)" \
               : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]) \
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]) : "memory")
template <int BITS> __device__ __forceinline__ void ld10(const void* const (&p)[10], uint4 (&v)[10]) {
  u4_t t[10];
  if constexpr (BITS == 3) LD10("sc0 sc1"); else if constexpr (BITS == 2) LD10("sc1"); else LD10("");
#pragma unroll
  for (int i = 0; i < 10; ++i) v[i] = make_uint4(t[i].x, t[i].y, t[i].z, t[i].w);
}
__device__ __forceinline__ unsigned poll(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the 16-byte word (8 fp16) at (row, 8-column chunk c8) of band `band` after seam s
__device__ __forceinline__ uint4 word_of(int s, int band, int row, int c8) {
  const unsigned a = (unsigned)(s * 0x9E3779B1u) ^ (unsigned)(band << 24) ^ (unsigned)(row << 12) ^ (unsigned)c8;
  return make_uint4(a, a * 3u + 1u, a ^ 0xA5A5A5A5u, a + 0x01234567u);
}

struct Stamps { long long publish, wait, first, all, total; };

// MODE 0 / 1 / 2 as above; PERMEMBER: flag per producer; grid = 256 blocks of 256 threads, 224 of them active
template <int MODE, bool PERMEMBER>
__global__ __launch_bounds__(256) void seam_kernel(uint16_t* __restrict__ buf0, uint16_t* __restrict__ buf1, unsigned* __restrict__ flags, int same_xcd,
                                                   long long* __restrict__ stamps, unsigned* __restrict__ bad) {
  const int tid = threadIdx.x;
  int band, member;
  if (same_xcd) {                      // XCD x = blockIdx % 8 hosts bands x and x + 8 (32 CUs per XCD = two bands)
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
    band = x + 8 * (slot >> 4); member = slot & 15;
  } else { band = blockIdx.x >> 4; member = blockIdx.x & 15; }
  if (band >= BANDS) return;
  unsigned* bflags = flags + band * 32;                  // [0] the band's counter, [16 + j] member j's flag (monotonic seam numbers)
  long long acc_pub = 0, acc_wait = 0, acc_first = 0, acc_all = 0;
  unsigned nbad = 0;
  const long long t_begin = now();
  for (int s = 1; s <= SEAMS; ++s) {
    uint16_t* buf = (s & 1) ? buf1 : buf0;               // double-buffered bands: seam s + 1 must not overwrite what a slow reader of seam s still reads
    uint16_t* bb = buf + (long)band * BAND_ELEMS;
    // ---- publish my 160 x 128 tile: 160 rows x 16 chunks of 16 B = 2 560 words, 10 per thread, row-contiguous 256-B segments
    const long long t0 = now();
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int w = tid + 256 * i, row = w >> 4, c = w & 15, c8 = member * 16 + c;
      const uint4 v = word_of(s, band, row, c8);
      void* dst = bb + (long)row * COLS + c8 * 8;
      if constexpr (MODE >= 5 && MODE <= 7) st_wt<3>(dst, v); else if constexpr (MODE >= 2 && MODE <= 4) st_wt<2>(dst, v); else *reinterpret_cast<uint4*>(dst) = v;
    }
    if constexpr (MODE == 1) {
      __threadfence();                                   // release at agent scope (L2 write-back)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (compiler hazard of the guide: never let the flag overtake the write-back)
    } else if constexpr (MODE >= 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                     // every thread's stores are out
    if (MODE != 0 && tid == 0) {
      if constexpr (PERMEMBER) __hip_atomic_store(bflags + 16 + member, (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(bflags, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t1 = now();
    // ---- wait + read.  The band is read as 16 slabs of 160 x 128 (one per producer), 10 words per thread per slab, 8 loads in flight.
    long long t2 = t1, t3 = t1;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (MODE != 0 && !PERMEMBER) {
      if (tid == 0) { while (poll(bflags) < (unsigned)(s * MEMBERS)) __builtin_amdgcn_s_sleep(1); }
      __syncthreads();
      if constexpr (MODE == 1 || MODE == 2 || MODE == 5) __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (agent scope on this target)
      t2 = now();
    }
    for (int j = 0; j < MEMBERS; ++j) {
      if (MODE != 0 && PERMEMBER) {
        if (tid == 0) { while (poll(bflags + 16 + j) < (unsigned)s) __builtin_amdgcn_s_sleep(1); }
        __syncthreads();
        if constexpr (MODE == 1 || MODE == 2 || MODE == 5) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (j == 0) t2 = now();
      }
      uint4 v[10];
      const void* src[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int w = tid + 256 * i, row = w >> 4, c8 = j * 16 + (w & 15);
        src[i] = bb + (long)row * COLS + c8 * 8;
      }
      if constexpr (MODE == 3 || MODE == 6) ld10<3>(src, v); else if constexpr (MODE == 4 || MODE == 7 || MODE == 8) ld10<2>(src, v); else ld10<0>(src, v);
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        if (MODE != 0) {
          const int w = tid + 256 * i, row = w >> 4, c8 = j * 16 + (w & 15);
          const uint4 e = word_of(s, band, row, c8);
          const unsigned nb = (v[i].x != e.x) + (v[i].y != e.y) + (v[i].z != e.z) + (v[i].w != e.w);
          nbad += nb;
          if (nb) atomicAdd(bad + 1 + ((j - member) & 15), 1u);
        }
        x.x ^= v[i].x; x.y ^= v[i].y; x.z ^= v[i].z; x.w ^= v[i].w;
      }
      if (j == 0) { asm volatile("" ::"v"(x.x)); t3 = now(); }      // first slab = the first two k-tiles of the next Linear are in registers
    }
    asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w));
    const long long t4 = now();
    acc_pub += t1 - t0; acc_wait += t2 - t1; acc_first += t3 - t2; acc_all += t4 - t2;
    __syncthreads();
  }
  const long long t_end = now();
  if (nbad) atomicAdd(bad, nbad);
  if (tid == 0) {
    long long* o = stamps + (long)blockIdx.x * 5;
    o[0] = acc_pub; o[1] = acc_wait; o[2] = acc_first; o[3] = acc_all; o[4] = t_end - t_begin;
  }
}

template <int MODE, bool PERMEMBER>
static void run(const char* name, int same_xcd, uint16_t* b0, uint16_t* b1, unsigned* flags, long long* stamps, unsigned* bad) {
  hipMemset(flags, 0, BANDS * 32 * sizeof(unsigned));
  hipMemset(bad, 0, 32 * sizeof(unsigned));
  hipMemset(stamps, 0, 256 * 5 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((seam_kernel<MODE, PERMEMBER>), dim3(256), dim3(256), 0, 0, b0, b1, flags, same_xcd, stamps, bad);
  hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess) { printf("%s: launch failed\n", name); return; }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(256 * 5);
  unsigned hbv[32];
  hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  hipMemcpy(hbv, bad, sizeof(hbv), hipMemcpyDeviceToHost);
  const unsigned hb = hbv[0];
  double s[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
  int n = 0;
  for (int b = 0; b < 256; ++b) {
    if (h[b * 5 + 4] == 0) continue;
    ++n;
    for (int k = 0; k < 5; ++k) { const double us = h[b * 5 + k] * 0.01 / SEAMS; s[k] += us; mx[k] = std::max(mx[k], us); }
  }
  for (int k = 0; k < 5; ++k) s[k] /= n;
  printf("%-36s %s | per seam, mean over %d workgroups (max): publish %5.2f (%5.2f)  wait %5.2f (%5.2f)  first slab %5.2f (%5.2f)  whole band %6.2f (%6.2f)  loop %6.2f (%6.2f) us | launch / %d seams = %6.2f us | bad words %u\n",
         name, same_xcd ? "one XCD per band" : "bands spread   ", n, s[0], mx[0], s[1], mx[1], s[2], mx[2], s[3], mx[3], s[4], mx[4], SEAMS, ms * 1000.0 / SEAMS, hb);
  if (hb) {
    printf("      bad 16-byte words by slab distance (j - member) mod 16:");
    for (int k = 0; k < 16; ++k) printf(" %u", hbv[1 + k]);
    printf("\n");
  }
}

int main() {
  uint16_t *b0, *b1; unsigned *flags, *bad; long long* stamps;
  const size_t bytes = (size_t)BANDS * BAND_ELEMS * 2;
  hipMalloc(&b0, bytes); hipMalloc(&b1, bytes); hipMalloc(&flags, BANDS * 32 * sizeof(unsigned)); hipMalloc(&bad, 32 * sizeof(unsigned)); hipMalloc(&stamps, 256 * 5 * sizeof(long long));
  hipMemset(b0, 0, bytes); hipMemset(b1, 0, bytes);
  printf("band-local seam: %d bands x %d workgroups, tile %d x %d fp16 (%d KiB published per workgroup, %d KiB read per workgroup), %d seams per launch\n", BANDS, MEMBERS, ROWS,
         TCOLS, ROWS * TCOLS * 2 / 1024, BAND_ELEMS * 2 / 1024, SEAMS);
  for (int rep = 0; rep < 1; ++rep) {
    for (int sx = 1; sx >= 0; --sx) {
      run<0, false>("mode 0 (no handshake)", sx, b0, b1, flags, stamps, bad);
      run<1, false>("mode 1 plain+release, band counter", sx, b0, b1, flags, stamps, bad);
      run<1, true>("mode 1 plain+release, member flags", sx, b0, b1, flags, stamps, bad);
      run<2, false>("mode 2 sc1 st, acquire, band counter", sx, b0, b1, flags, stamps, bad);
      run<2, true>("mode 2 sc1 st, acquire, member flags", sx, b0, b1, flags, stamps, bad);
      run<3, false>("mode 3 sc1 st, sc0sc1 ld, band cnt", sx, b0, b1, flags, stamps, bad);
      run<3, true>("mode 3 sc1 st, sc0sc1 ld, member fl", sx, b0, b1, flags, stamps, bad);
      run<4, false>("mode 4 sc1 st, sc1 ld, band counter", sx, b0, b1, flags, stamps, bad);
      run<4, true>("mode 4 sc1 st, sc1 ld, member flags", sx, b0, b1, flags, stamps, bad);
      run<5, true>("mode 5 sc0sc1 st, acquire, member fl", sx, b0, b1, flags, stamps, bad);
      run<7, true>("mode 7 sc0sc1 st, sc1 ld, member fl", sx, b0, b1, flags, stamps, bad);
      run<8, false>("mode 8 plain st, sc1 ld, band cnt", sx, b0, b1, flags, stamps, bad);
      run<8, true>("mode 8 plain st, sc1 ld, member fl", sx, b0, b1, flags, stamps, bad);
    }
    printf("\n");
  }
  return 0;
}
