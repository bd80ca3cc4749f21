// vt_gemm_f32r.hip — exact-fp32 GEMM / implicit conv1d for SMALL grids (the controller training step, fp32-mode Linears): 64 x 64 x 32
// block tile on v_mfma_f32_16x16x4_f32, operands HBM/L2 -> LDS by DMA through a RING of k-tiles with counted waits.
//
// Why a second fp32 kernel: the register-staged gemm_kernel (vt_gemm.hip) keeps ONE k-tile in flight per block.  A k-tile is 32 MFMAs
// per wave = 1024 cycles (0.45 us); a tile takes 1-2 us to arrive from L2 / HBM, so a block waits ~2/3 of the time unless 4 blocks
// share the CU — and the products of a training step (outputs of 64 ... 640 tiles, reductions of 512 ... 5120) put 2 blocks on a CU at
// best (tools/gemm_bench_f32.py: 45-64 TFLOP/s where the vendor library reaches 68-107).  Here a block keeps RING-1 = 3 k-tiles
// (48 KiB) in flight: `global_load_lds` 16 B per lane, a wave instruction fills 8 rows x 128 B, the XOR chunk swizzle applied to the
// SOURCE address (and again on the fragment read), `s_waitcnt vmcnt(N)` counting the younger tiles' DMA instructions, one raw
// s_barrier per k-tile.  Fragment reads and the MFMA order are those of gemm_kernel (Frag<float>: chunks 2g, 2g+1 of the row ->
// eight 4-deep MFMAs), so results are bit-identical to it for the same split-K factor.
// Conv mode (taps > 0, cin % 32 == 0): a k-tile lies inside one tap; rows whose tap falls outside [0, tin) read a zero page.
// Epilogue: + bias, fp32 row-major C (or the raw split-K slab).  Anything else (activation, column scale, residual, 16-bit types,
// K % 32 != 0) stays on gemm_kernel.
#include <stdlib.h>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_prof.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BM = 64, BN = 64, BK = 32;           // BK floats = 128 B per LDS row
constexpr int STAGE = (BM + BN) * 128;             // 16 KiB per k-tile

__device__ float vt_zero_page[32];                 // 128 B of zeros: the source of conv rows that fall in the padding

// RING = 4: 64 KiB, two blocks per CU.  (A 6-deep ring at one block per CU was measured on every shape of tools/gemm_bench_f32.py: never
// faster — a lone block is not waiting for data but for its own barrier / fragment-read / MFMA sequence, which a second block overlaps.)
// X3: split-bf16 arithmetic on the same fp32 tiles (vt_gemm.hip: a b ~= a_hi b_hi + a_lo b_hi + a_hi b_lo on the bf16 MFMA): the 8 floats of a
// fragment are exactly the 8 consecutive k of a v_mfma_f32_16x16x32_bf16 operand, so the split happens in registers after the fragment
// read (v_cvt_pk_bf16_f32 per pair) and a 32-deep k-tile costs 3 MFMAs per accumulator instead of 8.
__device__ __forceinline__ void split8(const Frag<float>& f, Frag<bf16_t>& hi, Frag<bf16_t>& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pk_bf16(f.v[2 * j], f.v[2 * j + 1]);
    l[j] = pk_bf16(f.v[2 * j] - __uint_as_float(h[j] << 16), f.v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
  }
  hi.v = __builtin_bit_cast(short8_t, make_uint4(h[0], h[1], h[2], h[3]));
  lo.v = __builtin_bit_cast(short8_t, make_uint4(l[0], l[1], l[2], l[3]));
}

template <int RING, bool X3>
__global__ __launch_bounds__(256, 2) void gemm_f32r_kernel(const VtGemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[RING * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;           // 2 x 2 waves, each 32 x 32
  const int g = lane >> 4, l15 = lane & 15;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int grp = blockIdx.z / p.splitk, slice = blockIdx.z - grp * p.splitk;
  const float* A = reinterpret_cast<const float*>(p.A) + (long)grp * p.a_gs;
  const float* W = reinterpret_cast<const float*>(p.W) + (long)grp * p.w_gs;

  const int nk_total = p.K / BK;
  const int nk_per = (nk_total + p.splitk - 1) / p.splitk;
  const int kt0 = slice * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);
  const int nk = max(0, kt1 - kt0);

  // DMA plan: a k-tile = 16 pieces of 1 KiB (8 rows x 128 B): pieces 0..7 = A rows, 8..15 = W rows; wave w issues pieces w, w+4, w+8, w+12.
  // lane -> (row inside the piece, chunk position); it fetches the chunk whose swizzled position is its own.
  const int r_in = lane >> 3, pch = lane & 7;
  const float* a_base[2];   // plain: row pointer; conv: sample base + (t*stride + off0) * lda (tap 0), may point outside: checked per tile
  int a_t[2];               // conv: t*stride + off0 of the row
  const float* w_src[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = (wave + 4 * e) * 8 + r_in;                     // tile row 0..63
    const int c = pch ^ ((r >> 1) & 7);
    const int m = min(m0 + r, p.M - 1);
    if (p.taps == 0) { a_base[e] = A + (long)m * p.lda + c * 4; a_t[e] = 0; }
    else {
      const int b = m / p.tout, t = m - b * p.tout;
      a_t[e] = t * p.stride + p.off0;
      a_base[e] = A + (long)b * p.tin * p.lda + c * 4;
    }
    w_src[e] = W + (long)min(n0 + r, p.N - 1) * p.ldw + c * 4;
  }
  const float* zero = vt_zero_page + (pch ^ 0) * 4;               // any 16-B chunk of the zero page
  auto stage = [&](int slot, int kt) {
    char* base = smem + slot * STAGE;
    const int k = kt * BK;
    int tap = 0, ci = k;
    if (p.taps != 0) { tap = k / p.cin; ci = k - tap * p.cin; }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float* src;
      if (p.taps == 0) src = a_base[e] + k;
      else {
        const int st = a_t[e] + tap * p.tstep;
        src = (st >= 0 && st < p.tin) ? a_base[e] + (long)st * p.lda + ci : zero;
      }
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(base + (wave + 4 * e) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_global_load_lds((glb_void*)(w_src[e] + k), (lds_void*)(base + BM * 128 + (wave + 4 * e) * 1024), 16, 0, 0);
  };
  // wait until k-tile i (of this block's nk) has landed — this wave's 4 pieces — leaving the younger tiles in flight; then the barrier
  auto arrive = [&](int i) {
    const int rem = min(RING - 3, nk - 1 - i);       // tiles younger than i already issued (4 DMA instructions of this wave each)
    if (rem >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (rem == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert(RING >= 4 && RING <= 6, "vmcnt table above covers up to 3 younger tiles");

  float4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  struct Frags { Frag<float> a[2], w[2]; };
  auto read_frags = [&](Frags& f, int slot) {
    const char* As = smem + slot * STAGE;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j) lds_frag(f.a[j], As, wm * 32 + j * 16 + l15, g);
#pragma unroll
    for (int n = 0; n < 2; ++n) lds_frag(f.w[n], Bs, wn * 32 + n * 16 + l15, g);
  };
  auto mfmas = [&](const Frags& f) {
    if constexpr (X3) {                   // small terms first, as gemm_kernel's split mode orders them
      Frag<bf16_t> ah[2], al[2], wh[2], wl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) split8(f.a[j], ah[j], al[j]);
#pragma unroll
      for (int n = 0; n < 2; ++n) split8(f.w[n], wh[n], wl[n]);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j) { mma16(acc[n][j], wl[n], ah[j]); mma16(acc[n][j], wh[n], al[j]); }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma16(acc[n][j], wh[n], ah[j]);
    } else {                              // the four accumulators advance together, one 4-deep slice at a time (each sums its slices in order)
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[n][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[n].v[e], f.a[j].v[e], acc[n][j], 0, 0, 0);
    }
  };
  // Software pipeline: the MFMAs of k-tile i only read registers, so they are issued LAST in iteration i and execute while the wave, in
  // iteration i+1, waits for tile i+2, passes the barrier, re-stages a slot and reads the next fragments.  Tile t lives in slot t % RING.
  // iteration i:  [tile i+1 landed + barrier]  [DMA tile i+RING-1 -> the slot of tile i-1, whose fragments every wave consumed before that
  // barrier]  [read tile i+1's fragments into the other register set]  [MFMAs of tile i]
  auto body = [&](int i, Frags& cur, Frags& nxt) {
    const bool more = i + 1 < nk;
    if (more) {
      arrive(i + 1);
      const int t = i + RING - 1;                                      // youngest tile not yet issued; its slot held tile i - 1
      if (t < nk) stage(t % RING, kt0 + t);
      read_frags(nxt, (i + 1) % RING);
    }
    mfmas(cur);
  };

#pragma unroll
  for (int i = 0; i < RING - 1; ++i)
    if (i < nk) stage(i, kt0 + i);
  Frags f0, f1;
  if (nk > 0) {
    {   // tile 0: wait for it alone (up to RING-2 younger tiles in flight)
      const int rem = min(RING - 2, nk - 1);
      if (rem >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (rem == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (rem == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    read_frags(f0, 0);
    for (int i = 0; i < nk; i += 2) {
      body(i, f0, f1);
      if (i + 1 < nk) body(i + 1, f1, f0);
    }
  }

  // ---------------- epilogue: lane holds C[m = .. + l15][n = .. + g*4 + r]
  const bool raw = p.splitk > 1;
  const bool vec = ((p.ldc & 3) == 0) && ((p.N & 3) == 0);
  const float* bias = (!raw && p.bias) ? p.bias + (long)grp * p.bias_gs : nullptr;
  float* Cb = reinterpret_cast<float*>(p.C) + (long)grp * p.c_gs + (raw ? (long)slice * p.c_slab : 0L);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 32 + j * 16 + l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = n0 + wn * 32 + i * 16 + g * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias[min(n + r, p.N - 1)];
      }
      float* C = Cb + (long)m * p.ldc + n;
      if (vec) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
      else
        for (int r = 0; r < 4; ++r) if (n + r < p.N) C[r] = v[r];
    }
  }
}

}  // namespace

// VLATOUCH_F32_RING=0 keeps every fp32 product on the register-staged kernel (A/B)
static int f32r_mode() { static const int m = [] { const char* e = getenv("VLATOUCH_F32_RING"); return e ? atoi(e) : -1; }(); return m; }

bool vt_gemm_f32r_eligible(const VtGemmParams& p) {
  if (f32r_mode() == 0) return false;
  if (p.a_dtype != VT_F32 || (p.w_dtype != VT_F32 && p.w_dtype != VT_F32X3) || p.c_dtype != VT_F32) return false;
  if (p.w_dtype == VT_F32X3 && (f32r_mode() == 1 || p.K % 64)) return false;     // VLATOUCH_F32_RING=1: ring for exact fp32 only (A/B); K % 64: gemm_kernel's slices
  if (p.act != VT_ACT_NONE || p.colscale || p.residual || p.hn_w0 || p.hn_w1 || p.cmap) return false;
  if (p.K % BK || p.K < 2 * BK || p.lda % 4 || p.ldw % 4) return false;
  if (p.taps && (p.cin % BK || p.K != p.taps * p.cin)) return false;
  if (p.M < 32) return false;                                   // tiny-M products keep the 32 x 64 configuration of gemm_kernel
  // large grids are MFMA-bound on the 128 x 128 tile of gemm_kernel (121 TFLOP/s at 4096^3); the ring pays off when few blocks share a CU
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.groups * p.splitk;
  return tiles128 < 1024;
}

int vt_gemm_f32r_launch(const VtGemmParams& p, hipStream_t s) {
  VtProfScope prof(5, p, s);
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.groups * p.splitk);
  if (p.w_dtype == VT_F32X3) hipLaunchKernelGGL((gemm_f32r_kernel<4, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_f32r_kernel<4, false>), grid, dim3(256), 0, s, p);
  return vt_check_launch();
}
