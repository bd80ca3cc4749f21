// vt_gemm_pws.hip — small-M companion of vt_gemm_pw.hip: 16-bit Linears with FROZEN, fragment-packed weights at M <= 512 rows (the denoise loop
// of RDT at batch 1..7: M = batch x 67; the robot loop of the reference runs batch 1, residual_controller/frank_inference_eef.py:495-533).
//
// What bounds a Linear at M = 67: its 8.4 MB of weights have to cross HBM once (1.7 us at 5 TB/s) and a CU ingests ~40 B/clk, so >= 100 CUs
// must pull in parallel and every byte should enter a CU once.  The generic path (64 x 64 register-staged tiles, split-K 8..16 into fp32
// slabs + a slab-reduction kernel) took 12.6 + 5 us per Linear.  Here:
//   * grid = m-tiles (96 rows) x (N / 64) column blocks x S k-slices, S in {1, 2, 4, 8} chosen so that >= 128 blocks exist;
//   * a block (4 waves) owns 96 x 64 outputs over its k-slice: wave (c, p) takes the 32 columns c and the k-tiles of parity p.  Weights come
//     straight from the fragment-packed copy (vt_pack_w32) into registers, 1 KiB per load; the activation rows go HBM/L2 -> LDS by DMA
//     (12 pieces of 8 rows per k-tile, XOR-swizzled on the source address) and are shared by the two column halves;
//   * the k-slice moves in chunks of 4 k-tiles (48 KiB of A + 32 KiB of W per block), two chunks in flight (double-buffered LDS and weight
//     registers), counted `s_waitcnt vmcnt(20)` + raw s_barrier;
//   * the two k-parity waves of a column half add through LDS; with S > 1 the block's fp32 partial goes to slab[slice] (plain stores, agent-
//     scope release, ticket); the LAST block of a tile to arrive sums the S slabs in slice order (deterministic: independent of arrival order)
//     and runs the shared epilogue (bias / per-head RMSNorm / activation / column scale / residual) — no second kernel, no extra boundary.
//     The ticket counters are self-resetting; the RDT driver zeroes them once per call.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_gemm_epilogue.h"
#include "vt_prof.h"
#include "vt_host.h"
#include "../../include/vlatouch.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) int int4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

constexpr int BM = 96, BN = 64, BK = 64;
constexpr int TMW = BM / 32;                     // 32-row MFMA tiles per wave
constexpr int A_TILE = BM * 128;                 // one k-tile of the activation rows (12 KiB)
constexpr int CH = 4;                            // k-tiles per chunk
constexpr int CHUNK = CH * A_TILE;               // 48 KiB
constexpr int OPS = CH * 3 + 8;                  // vector-memory operations per wave per chunk: 12 DMA pieces + 8 weight loads

template <typename T16> __device__ __forceinline__ float16_t mma32s(const int4_t w, const int4_t a, const float16_t c);
template <> __device__ __forceinline__ float16_t mma32s<bf16_t>(const int4_t w, const int4_t a, const float16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, a), c, 0, 0, 0);
}
template <> __device__ __forceinline__ float16_t mma32s<half_t>(const int4_t w, const int4_t a, const float16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, w), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
}
// (A/B: -DVLATOUCH_PWS_NT puts the nt cache policy on the weight stream — every fragment is read by ONE block per launch)
#ifdef VLATOUCH_PWS_NT
#define VT_PWS_POL " nt"
#else
#define VT_PWS_POL ""
#endif
template <int OFF, bool FIRST>
__device__ __forceinline__ void pws_wload(int4_t& d, const unsigned voff, const char* sbase) {
  if constexpr (FIRST) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" VT_PWS_POL : "=v"(d) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" VT_PWS_POL : "=v"(d) : "v"(voff), "s"(sbase), "i"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void pws_wait(int4_t (&w)[8]) {
#ifdef VLATOUCH_DRAIN_WAITS      // debug build (tools/drain_waits_check.sh)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : [n] "i"(N) : "memory");
#endif
}

template <typename T16, typename TC>
__global__ __launch_bounds__(256, 1) void gemm_pws_kernel(const VtGemmParams p, const int S, const int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * CHUNK];             // two chunks of A; later the reduction / output tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wp = wave >> 1;          // column half, k-tile parity
  const int l31 = lane & 31, hk = lane >> 5;
  const int slice = blockIdx.x % S;
  const int tile_id = blockIdx.x / S;
  const int tn = tile_id % tiles_n, tm = tile_id / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = p.K / BK / S;                     // k-tiles of this slice (multiple of CH)
  const int kt0 = slice * nkt;
  const int NC = nkt / CH;

  const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
  int asrc[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int r = (wave + 4 * e) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    asrc[e] = (int)(((long)(min(m0 + r, p.M - 1) - m0) * p.lda + c * 8) * 2);
  }
  const char* wbase = reinterpret_cast<const char*>(p.Wp) + ((long)(n0 / 32 + wc) * (p.K / 16)) * 1024;
  wbase = reinterpret_cast<const char*>(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long)wbase >> 32)) << 32) |
                                        (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long)wbase));
  const unsigned wvoff = lane * 16;

  int4_t wb[2][8];                                  // [chunk buffer][own k-tile 0/1 of the chunk x 4 k-steps]
  auto issue = [&](const int c, auto bc) {          // chunk c of the slice -> LDS buffer b, weight buffer b
    constexpr int b = decltype(bc)::value;
#pragma unroll
    for (int kk = 0; kk < CH; ++kk)
#pragma unroll
      for (int e = 0; e < 3; ++e)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(smem + b * CHUNK + kk * A_TILE + (wave + 4 * e) * 1024), 16, asrc[e],
                                                 (kt0 + c * CH + kk) * (BK * 2), 0, 0);
    const char* sb0 = wbase + (long)(kt0 + c * CH + wp) * 4096;         // this wave's k-tiles of the chunk: wp and wp + 2
    pws_wload<0, true>(wb[b][0], wvoff, sb0);
    pws_wload<1024, false>(wb[b][1], wvoff, sb0);
    pws_wload<2048, false>(wb[b][2], wvoff, sb0);
    pws_wload<3072, false>(wb[b][3], wvoff, sb0);
    const char* sb1 = sb0 + 2 * 4096;
    pws_wload<0, true>(wb[b][4], wvoff, sb1);
    pws_wload<1024, false>(wb[b][5], wvoff, sb1);
    pws_wload<2048, false>(wb[b][6], wvoff, sb1);
    pws_wload<3072, false>(wb[b][7], wvoff, sb1);
  };

  const int xr = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = l31 * 128 + (((s * 2 + hk) ^ xr) * 16);

  float16_t acc[TMW];
#pragma unroll
  for (int j = 0; j < TMW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  issue(0, std::integral_constant<int, 0>{});
  if (NC > 1) issue(1, std::integral_constant<int, 1>{});

  auto body = [&](const int c, auto bc) {
    constexpr int b = decltype(bc)::value;
    // ONE statement names the weight registers (a second one in another branch makes hipcc route the values through copies that it
    // places BEFORE the wait of one branch — i.e. it copies registers whose loads have not landed); the last chunk drains the queue with
    // an operand-less wait ahead of it
    if (c + 1 >= NC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pws_wait<OPS>(wb[b]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                   // every wave's pieces of chunk c are block-visible
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int4_t af[TMW];
#pragma unroll
        for (int j = 0; j < TMW; ++j) af[j] = *reinterpret_cast<const int4_t*>(smem + b * CHUNK + (wp + 2 * kk) * A_TILE + j * 4096 + foff[s]);
#pragma unroll
        for (int j = 0; j < TMW; ++j) acc[j] = mma32s<T16>(wb[b][kk * 4 + s], af[j], acc[j]);
      }
    if (c + 2 < NC) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                 // every wave is done reading buffer b
      __builtin_amdgcn_sched_barrier(0);
      issue(c + 2, bc);
    }
  };
  for (int c = 0; c < NC; c += 2) {
    body(c, std::integral_constant<int, 0>{});
    if (c + 1 < NC) body(c + 1, std::integral_constant<int, 1>{});
  }
  __syncthreads();                                  // all fragment reads done: the buffers become scratch

  // ---------------- the two k-parity waves of a column half add through LDS (lane-contiguous: conflict-free)
  float* xch = reinterpret_cast<float*>(smem) + (long)wc * (TMW * 16) * 64 + lane;
  if (wp == 1) {
#pragma unroll
    for (int j = 0; j < TMW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 64] = acc[j][r];
  }
  __syncthreads();
  float* tile = reinterpret_cast<float*>(smem + 2 * TMW * 16 * 64 * 4);          // [96][64] fp32 beyond the exchange area, 16-byte columns swizzled by the row
  if (wp == 0) {
#pragma unroll
    for (int j = 0; j < TMW; ++j) {
      const int m = j * 32 + l31;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float4 v;
        v.x = acc[j][rq * 4 + 0] + xch[(j * 16 + rq * 4 + 0) * 64];
        v.y = acc[j][rq * 4 + 1] + xch[(j * 16 + rq * 4 + 1) * 64];
        v.z = acc[j][rq * 4 + 2] + xch[(j * 16 + rq * 4 + 2) * 64];
        v.w = acc[j][rq * 4 + 3] + xch[(j * 16 + rq * 4 + 3) * 64];
        const int n4 = wc * 8 + rq * 2 + hk;
        *reinterpret_cast<float4*>(tile + m * BN + ((n4 ^ (m & 7)) * 4)) = v;
      }
    }
  }
  __syncthreads();

  // ---------------- row-contiguous side: 16 lanes cover the 64 columns (one head) of a row, 16 rows per pass
  const int c4 = lane & 15, rsub = tid >> 4;
  const int n = n0 + c4 * 4;
  const bool col_ok = n < p.N;
  int* flag = reinterpret_cast<int*>(smem);          // the exchange area is free again
  if (p.splitk > 1) {
    // the generic kernel's split-K contract (vt_gemm.h): raw fp32 partial slab of slice `slice` at C + slice * c_slab, no epilogue; the
    // caller's slab-reduction kernel (which also carries the Linear's epilogue and, for residual Linears, the RMSNorm that follows)
    // combines them behind the kernel boundary — on this chip an in-launch combine costs more than that boundary (5-13 us per seam)
    float* slab = reinterpret_cast<float*>(p.C) + (long)slice * p.c_slab;
#pragma unroll
    for (int it = 0; it < BM / 16; ++it) {
      const int row = it * 16 + rsub, m = m0 + row;
      if (m < p.M && col_ok) *reinterpret_cast<float4*>(slab + (long)m * p.ldc + n) = *reinterpret_cast<const float4*>(tile + row * BN + ((c4 ^ (row & 7)) * 4));
    }
    return;
  }
  if (S > 1) {
    float* slab = reinterpret_cast<float*>(p.sk_ws) + (long)slice * p.M * p.N;
#pragma unroll
    for (int it = 0; it < BM / 16; ++it) {
      const int row = it * 16 + rsub, m = m0 + row;
      if (m < p.M && col_ok) *reinterpret_cast<float4*>(slab + (long)m * p.N + n) = *reinterpret_cast<const float4*>(tile + row * BN + ((c4 ^ (row & 7)) * 4));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(p.sk_cnt + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == S - 1;
      if (last) {
        __hip_atomic_store(p.sk_cnt + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // self-resetting: the next launch finds zero
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      flag[0] = last;
    }
    __syncthreads();
    if (!flag[0]) return;
  }
  const float* bias = p.bias;
  const float* hw = nullptr;
  if (p.hn_w0 && n0 < p.hn_c0_end) hw = p.hn_w0;
  else if (p.hn_w1 && n0 >= p.hn_c0_end && n0 < p.hn_c1_end) hw = p.hn_w1;
  TC* Cg = reinterpret_cast<TC*>(p.C);
  const TC* Rg = reinterpret_cast<const TC*>(p.residual);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 b4 = (bias && col_ok) ? *reinterpret_cast<const float4*>(bias + n) : zero4;
  const float4 cs4 = (p.colscale && col_ok) ? *reinterpret_cast<const float4*>(p.colscale + n) : one4;
  const float4 hw4 = hw ? *reinterpret_cast<const float4*>(hw + c4 * 4) : one4;
#pragma unroll 2
  for (int it = 0; it < BM / 16; ++it) {
    const int row = it * 16 + rsub, m = m0 + row;
    float4 x;
    if (S > 1) {                                       // slabs in slice order (this block's own partial is re-read: L2-resident)
      x = zero4;
      if (m < p.M && col_ok) {
        const float* sp = reinterpret_cast<const float*>(p.sk_ws) + (long)m * p.N + n;
        for (int s = 0; s < S; ++s) {
          const float4 v = *reinterpret_cast<const float4*>(sp + (long)s * p.M * p.N);
          x.x += v.x; x.y += v.y; x.z += v.z; x.w += v.w;
        }
      }
    } else {
      x = *reinterpret_cast<const float4*>(tile + row * BN + ((c4 ^ (row & 7)) * 4));
    }
    if (p.act != VT_ACT_NONE) vt_epi_segment<TC, 0, true>(p, x, b4, cs4, hw, hw4, Cg, Rg, m, n, n0, col_ok);
    else vt_epi_segment<TC, 0, false>(p, x, b4, cs4, hw, hw4, Cg, Rg, m, n, n0, col_ok);
  }
}

}  // namespace

static int g_vt_pws_on = 1;      // VLATOUCH_PWS=0 / vt_tune(3, 0) disables the kernel; vt_tune(4, S) forces the split factor (0 = choose)
static int g_vt_pws_s = 0;

void vt_gemm_pws_tune(int knob, int value) {
  if (knob == 3) g_vt_pws_on = value != 0;
  if (knob == 4) g_vt_pws_s = value;
}

static int pws_split(const VtGemmParams& p) {
  if (p.splitk > 1) return p.splitk;
  const long tiles = (long)((p.M + BM - 1) / BM) * (p.N / BN);
  int S = 1;
  if (g_vt_pws_s < 0) while (tiles * S < 128 && S < 8 && (p.K / BK) % (2 * S * CH) == 0) S *= 2;      // knob 4 = -1: choose; default: no in-launch combine
  if (g_vt_pws_s > 0 && (p.K / BK) % (g_vt_pws_s * CH) == 0) S = g_vt_pws_s;
  while (S > 1 && (!p.sk_ws || !p.sk_cnt || (size_t)S * p.M * p.N * 4 > p.sk_ws_bytes || tiles > p.sk_cnt_n)) S >>= 1;
  return S;
}

bool vt_gemm_pws_eligible(const VtGemmParams& p) {
  static const bool init = [] { const char* e = getenv("VLATOUCH_PWS"); if (e) g_vt_pws_on = atoi(e) != 0; return true; }();
  (void)init;
  if (!g_vt_pws_on || !p.Wp || p.cmap || p.groups != 1 || p.taps != 0 || p.splitk < 1) return false;
  if (p.splitk > 1 && (p.c_dtype != VT_F32 || (p.K / BK) % (p.splitk * CH) || p.ldc % 4)) return false;     // slab mode: whole chunks per slice
  if ((p.a_dtype != VT_BF16 && p.a_dtype != VT_F16) || p.w_dtype != p.a_dtype) return false;
  if (p.c_dtype != p.a_dtype && p.c_dtype != VT_F32) return false;
  if (p.M < 1 || p.M > 512 || p.N % BN || p.K % (CH * BK) || p.lda % 8 || p.lda >= (1 << 21) || p.ldc % 4 || (p.residual && p.ldr % 4)) return false;
  if ((p.hn_w0 || p.hn_w1) && (p.hn_c0_end % 64 || p.hn_c1_end % 64)) return false;
  return true;
}

int vt_gemm_pws_launch(const VtGemmParams& p, hipStream_t s) {
  const int S = pws_split(p);
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  VtProfScope prof(3, p, s);
  const bool c16 = p.c_dtype != VT_F32;
#define VT_PWS_GO(T16, TC) hipLaunchKernelGGL((gemm_pws_kernel<T16, TC>), dim3(tiles_n * tiles_m * S), dim3(256), 0, s, p, S, tiles_n)
  if (p.a_dtype == VT_BF16) { if (c16) VT_PWS_GO(bf16_t, bf16_t); else VT_PWS_GO(bf16_t, float); }
  else { if (c16) VT_PWS_GO(half_t, half_t); else VT_PWS_GO(half_t, float); }
#undef VT_PWS_GO
  return vt_check_launch();
}
