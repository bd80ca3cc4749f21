// vt_gemm.hip — generic MFMA GEMM / implicit-conv1d kernel for gfx950 (see vt_gemm.h for semantics).
//
// Structure (v1, register-staged): 256 threads = 4 waves; block tile BM x BN x BK with BK = 128 bytes of
// the compute type per row (64 bf16 / 32 f32).  Global -> registers (16 B per thread per pass, coalesced
// 128-B row segments) -> LDS rows of 8 XOR-swizzled 16-B chunks (conflict-free ds_read_b128 for the
// 16-row fragment reads) -> v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32.  The NEXT k-tile's global
// loads are issued before the current tile's MFMAs (software prefetch through registers).
// Operands are swapped (D = W_tile * A_tile^T) so each lane ends with 4 CONSECUTIVE n for one m:
// the epilogue reads bias/colscale/residual and writes C with 16-B (f32) / 8-B (bf16) vectors.
#include <type_traits>
#include "vt_common.h"
#include "vt_gemm.h"
#include "vt_prof.h"

VtProfState g_vt_prof;

namespace {

template <typename TA, typename TW> struct ChunkLoad;
// same type: one 16-B load
template <typename T> struct ChunkLoad<T, T> {
  __device__ static __forceinline__ uint4 load(const T* p) { return *reinterpret_cast<const uint4*>(p); }
};
// f32 activations feeding a bf16 MFMA: 8 floats -> 8 bf16
template <> struct ChunkLoad<float, bf16_t> {
  __device__ static __forceinline__ uint4 load(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    uint4 r;
    r.x = (uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16);
    r.y = (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16);
    r.z = (uint32_t)f2bf(b.x) | ((uint32_t)f2bf(b.y) << 16);
    r.w = (uint32_t)f2bf(b.z) | ((uint32_t)f2bf(b.w) << 16);
    return r;
  }
};

// split-bf16 mode ("bf16x3"): A and W are fp32 in global; each value is split at staging time into
// hi = bf16(x), lo = bf16(x - hi) and the product is accumulated as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo on the
// bf16 MFMA pipe (relative product error ~2^-16 instead of ~2^-8): near-fp32 accuracy at 1/3 of the bf16 rate
// (vs 1/16 for the fp32 MFMA).  Used by the interpolant U-Nets, whose score output is amplified by the SDE.
struct x3_t {};
template <typename TW> struct Cmp { using type = TW; using storage = TW; static constexpr bool X3 = false; };
template <> struct Cmp<x3_t> { using type = bf16_t; using storage = float; static constexpr bool X3 = true; };

__device__ __forceinline__ void split8(const float* p, uint4& hi, uint4& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bf16_t hh = f2bf(f[j]);
    h[j] = hh;
    l[j] = f2bf(f[j] - bf2f(hh));
  }
  hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// the same split on 8 floats already in registers (two raw 16-B loads): hi = bf16(x), lo = bf16(x - hi), packed pairwise
__device__ __forceinline__ void split8r(const uint4& a, const uint4& b, uint4& hi, uint4& lo) {
  const float f[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                      __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = pk_bf16(f[2 * j], f[2 * j + 1]);
    const float r0 = f[2 * j] - __uint_as_float(h[j] << 16), r1 = f[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u);
    l[j] = pk_bf16(r0, r1);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <typename TC> struct Store4;
template <> struct Store4<float> {
  __device__ static __forceinline__ void st(float* p, const float v[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
  __device__ static __forceinline__ void ld(const float* p, float v[4]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
};
template <> struct Store4<bf16_t> {
  __device__ static __forceinline__ void st(bf16_t* p, const float v[4]) {
    uint2 t;
    t.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    t.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = t;
  }
  __device__ static __forceinline__ void ld(const bf16_t* p, float v[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
  }
};

template <> struct Store4<half_t> {
  __device__ static __forceinline__ void st(half_t* p, const float v[4]) {
    half_t t[4] = {f2h(v[0]), f2h(v[1]), f2h(v[2]), f2h(v[3])};
    *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(t);
  }
  __device__ static __forceinline__ void ld(const half_t* p, float v[4]) {
    half_t t[4];
    *reinterpret_cast<uint2*>(t) = *reinterpret_cast<const uint2*>(p);
    v[0] = h2f(t[0]); v[1] = h2f(t[1]); v[2] = h2f(t[2]); v[3] = h2f(t[3]);
  }
};

template <typename TA, typename TW, typename TC, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, (TM * TN >= 16 ? 2 : 1)) void gemm_kernel(const VtGemmParams p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  using TCmp = typename Cmp<TW>::type;       // MFMA operand type
  using TWS = typename Cmp<TW>::storage;     // W element type in global memory
  constexpr bool X3 = Cmp<TW>::X3;
  constexpr int EPC = Elem<TCmp>::EPC;      // k elements per 16-B LDS chunk of the compute type
  constexpr int BK = 8 * EPC;               // 128 B per LDS row
  constexpr int KSTEPS = BK / 32;           // 32-deep MFMA steps per tile
  constexpr int PA = BM / 32, PB = BN / 32; // staging passes (32 rows x 8 chunks per pass)
  constexpr int NPL = X3 ? 2 : 1;           // LDS planes (hi, lo)
  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * 128 * NPL];
  char* As = smem;
  char* Bs = smem + BM * 128;
  char* As2 = smem + (BM + BN) * 128;       // lo planes (split mode only)
  char* Bs2 = As2 + BM * 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = lane >> 4, l15 = lane & 15;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int grp = blockIdx.z / p.splitk, slice = blockIdx.z - grp * p.splitk;

  const TA* A = reinterpret_cast<const TA*>(p.A) + (long)grp * p.a_gs;
  const TWS* W = reinterpret_cast<const TWS*>(p.W) + (long)grp * p.w_gs;

  const int nk_total = (p.K + BK - 1) / BK;
  const int nk_per = (nk_total + p.splitk - 1) / p.splitk;
  const int kt0 = slice * nk_per;
  const int kt1 = min(nk_total, kt0 + nk_per);

  // per-thread staging coordinates
  const int cr = tid >> 3, cc = tid & 7;
  long a_row_off[PA];   // plain: m*lda ; conv: b*tin*lda (row base of the sample)
  int a_t[PA];          // conv: t*stride + off0 ; plain: 0
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int m = m0 + i * 32 + cr;
    a_ok[i] = m < p.M;
    if (p.taps == 0) {
      a_row_off[i] = (long)m * p.lda;
      a_t[i] = 0;
    } else {
      const int b = m / p.tout, t = m - b * p.tout;
      a_row_off[i] = (long)b * p.tin * p.lda;
      a_t[i] = t * p.stride + p.off0;
    }
  }
  long w_row_off[PB];
  bool w_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int n = n0 + i * 32 + cr;
    w_ok[i] = n < p.N;
    w_row_off[i] = (long)n * p.ldw;
  }

  uint4 ra[PA], rb[PB], ra2[X3 ? PA : 1], rb2[X3 ? PB : 1];
  auto load_tile = [&](int kt) {
    const int k = kt * BK + cc * EPC;
    const bool kin = k < p.K;
    int tap = 0, c = k;
    if (p.taps != 0) { tap = k / p.cin; c = k - tap * p.cin; }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0), v2 = make_uint4(0, 0, 0, 0);
      const TA* src = nullptr;
      if (a_ok[i] && kin) {
        if (p.taps == 0) {
          src = A + a_row_off[i] + k;
        } else {
          const int st = a_t[i] + tap * p.tstep;
          if (st >= 0 && st < p.tin) src = A + a_row_off[i] + (long)st * p.lda + c;
        }
      }
      if (src) {
        if constexpr (X3) {      // raw fp32 now, hi/lo split in store_tile: splitting here would wait for the load before the MFMAs
          v = *reinterpret_cast<const uint4*>(src);
          v2 = *reinterpret_cast<const uint4*>(src + 4);
        } else v = ChunkLoad<TA, TCmp>::load(src);
      }
      ra[i] = v;
      if constexpr (X3) ra2[i] = v2;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0), v2 = make_uint4(0, 0, 0, 0);
      if (w_ok[i] && kin) {
        if constexpr (X3) {
          v = *reinterpret_cast<const uint4*>(W + w_row_off[i] + k);
          v2 = *reinterpret_cast<const uint4*>(W + w_row_off[i] + k + 4);
        } else v = *reinterpret_cast<const uint4*>(W + w_row_off[i] + k);
      }
      rb[i] = v;
      if constexpr (X3) rb2[i] = v2;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int r = i * 32 + cr;
      if constexpr (X3) {
        uint4 hi, lo;
        split8r(ra[i], ra2[i], hi, lo);
        *reinterpret_cast<uint4*>(As + r * 128 + swz(r, cc) * 16) = hi;
        *reinterpret_cast<uint4*>(As2 + r * 128 + swz(r, cc) * 16) = lo;
      } else {
        *reinterpret_cast<uint4*>(As + r * 128 + swz(r, cc) * 16) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int r = i * 32 + cr;
      if constexpr (X3) {
        uint4 hi, lo;
        split8r(rb[i], rb2[i], hi, lo);
        *reinterpret_cast<uint4*>(Bs + r * 128 + swz(r, cc) * 16) = hi;
        *reinterpret_cast<uint4*>(Bs2 + r * 128 + swz(r, cc) * 16) = lo;
      } else {
        *reinterpret_cast<uint4*>(Bs + r * 128 + swz(r, cc) * 16) = rb[i];
      }
    }
  };

  float4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  if (kt0 < kt1) {
    load_tile(kt0);
    store_tile();
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = kt + 1 < kt1;
      if (more) load_tile(kt + 1);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        Frag<TCmp> af[TM], wf[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) lds_frag(af[j], As, wm * TM * 16 + j * 16 + l15, ks * 4 + g);
#pragma unroll
        for (int i = 0; i < TN; ++i) lds_frag(wf[i], Bs, wn * TN * 16 + i * 16 + l15, ks * 4 + g);
        if constexpr (X3) {
          Frag<TCmp> af2[TM], wf2[TN];
#pragma unroll
          for (int j = 0; j < TM; ++j) lds_frag(af2[j], As2, wm * TM * 16 + j * 16 + l15, ks * 4 + g);
#pragma unroll
          for (int i = 0; i < TN; ++i) lds_frag(wf2[i], Bs2, wn * TN * 16 + i * 16 + l15, ks * 4 + g);
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {   // small terms first
              mma16(acc[i][j], wf2[i], af[j]);
              mma16(acc[i][j], wf[i], af2[j]);
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) mma16(acc[i][j], wf[i], af[j]);
      }
      __syncthreads();
      if (more) {
        store_tile();
        __syncthreads();
      }
    }
  }

  // ---------------- epilogue: lane holds C[m = .. + l15][n = .. + g*4 + r]
  const bool raw = p.splitk > 1;
  const bool vec = ((p.ldc & 3) == 0) && ((p.N & 3) == 0) && (p.residual == nullptr || (p.ldr & 3) == 0);
  const float* bias = p.bias ? p.bias + (long)grp * p.bias_gs : nullptr;
  const float* cs = p.colscale;
  // two copies of the store loop: the activation math inlined at every accumulator register is large, and jumping over it
  // register by register costs an instruction-cache miss each time; the common case (no activation) gets a compact copy
  auto store_all = [&](auto with_act) {
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * TM * 16 + j * 16 + l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wn * TN * 16 + i * 16 + g * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (raw) {
        float* C = reinterpret_cast<float*>(p.C) + (long)grp * p.c_gs + (long)slice * p.c_slab + (long)m * p.ldc + n;
        if (vec) Store4<float>::st(C, v);
        else
          for (int r = 0; r < 4; ++r) if (n + r < p.N) C[r] = v[r];
        continue;
      }
      TC* C = reinterpret_cast<TC*>(p.C) + (long)grp * p.c_gs + (long)m * p.ldc + n;
      const TC* R = p.residual ? reinterpret_cast<const TC*>(p.residual) + (long)grp * p.r_gs + (long)m * p.ldr + n : nullptr;
      float rv[4] = {0.f, 0.f, 0.f, 0.f};
      if (R) {
        if (vec) Store4<TC>::ld(R, rv);
        else
          for (int r = 0; r < 4; ++r) if (n + r < p.N) rv[r] = Elem<TC>::to_f(R[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = min(n + r, p.N - 1);
        float x = v[r];
        if (bias) x += bias[nn];
        if constexpr (decltype(with_act)::value) x = act_apply(x, p.act);
        if (cs) x *= cs[nn];
        v[r] = x + rv[r];
      }
      if (vec) Store4<TC>::st(C, v);
      else
        for (int r = 0; r < 4; ++r) if (n + r < p.N) C[r] = Elem<TC>::from_f(v[r]);
    }
  }
  };
  if (p.act == VT_ACT_NONE || raw) store_all(std::false_type{});
  else store_all(std::true_type{});
}

template <typename TA, typename TW, typename TC>
int launch_cfg(const VtGemmParams& p, hipStream_t s) {
  const int z = p.groups * p.splitk;
  static const bool trace = getenv("VLATOUCH_GEMM_TRACE") != nullptr;      // one line per launch of the register-staged kernel (which shapes still land here?)
  if (trace) fprintf(stderr, "[vt_gemm generic] M=%d N=%d K=%d groups=%d splitk=%d taps=%d a=%d w=%d c=%d act=%d\n", p.M, p.N, p.K, p.groups, p.splitk, p.taps, p.a_dtype, p.w_dtype, p.c_dtype, p.act);
  VtProfScope prof(5, p, s);   // class 5: this register-staged kernel (exact-fp32 / split-bf16 / small-M products; the training step's GEMMs)
  // tile choice: enough 128x128 tiles to fill the chip -> large tile; tiny M -> 32x64; else 64x64
  const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * z;
  if constexpr (!Cmp<TW>::X3) {
    if (p.M >= 128 && tiles128 >= 192) {
      dim3 grid((p.N + 127) / 128, (p.M + 127) / 128, z);
      hipLaunchKernelGGL((gemm_kernel<TA, TW, TC, 2, 2, 4, 4>), grid, dim3(256), 0, s, p);
      return vt_check_launch();
    }
  }
  if (p.M <= 32) {
    dim3 grid((p.N + 63) / 64, (p.M + 31) / 32, z);
    hipLaunchKernelGGL((gemm_kernel<TA, TW, TC, 1, 4, 2, 1>), grid, dim3(256), 0, s, p);
  } else {
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, z);
    hipLaunchKernelGGL((gemm_kernel<TA, TW, TC, 2, 2, 2, 2>), grid, dim3(256), 0, s, p);
  }
  return vt_check_launch();
}

}  // namespace

// Host entry used by every driver in the library (and exported through vt_gemm in vt_api.hip).
int vt_gemm_launch(const VtGemmParams& p, hipStream_t s) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return VT_ERR_ARG;
  const int epc = p.w_dtype == VT_F32 ? 4 : 8;   // k elements per staged chunk (bf16 and split-bf16: 8)
  const int epa = epc;  // A chunks are loaded in units of the compute type's chunk
  if (p.K % epc || p.ldw % epc || p.lda % epa) return VT_ERR_ARG;      // 16-B chunk granularity
  if (p.taps && (p.cin % epc || p.K != p.taps * p.cin)) return VT_ERR_ARG;
  if (p.splitk < 1 || p.groups < 1) return VT_ERR_ARG;
  if (p.splitk > 1 && p.c_dtype != VT_F32) return VT_ERR_ARG;
  if (p.pf_ptr && p.pf_bytes >= (1ul << 31)) return VT_ERR_ARG;      // prefetch hint: 32-bit byte arithmetic in the kernel
  if (p.xn_out || p.rs_part) {        // fused RMSNorm hand-off: only the weights-in-registers tile implements it, and the caller has checked that it takes this shape
    if (p.groups != 1 || (p.xn_out && (!p.xn_gain || !p.xn_part || p.c_dtype != VT_F32 || !p.residual || p.act != VT_ACT_NONE || p.hn_w0 || p.hn_w1 || p.xn_ld % 4)) ||
        (p.rs_part && (p.rs_n < 4 || p.rs_n > 32 || p.rs_n % 4)) || !vt_gemm_fast_eligible(p) || !vt_gemm_pw_eligible(p))
      return VT_ERR_UNSUPPORTED;
    return vt_gemm_pw_launch(p, s);
  }
  if (vt_gemm_pws_eligible(p) && (p.M <= 192 || !vt_gemm_fast_eligible(p))) return vt_gemm_pws_launch(p, s);   // small M, frozen packed weights
  if (vt_gemm_fast_eligible(p)) return vt_gemm_fast_launch(p, s);     // large bf16 GEMMs: LDS-DMA pipeline (vt_gemm_fast.hip)
  if (p.hn_w0 || p.hn_w1 || p.cmap) return VT_ERR_UNSUPPORTED;         // fused head-norm / tile-stream output exist only on the fast path
  if (p.a_dtype == VT_BF16 && p.w_dtype == VT_BF16) {
    if (p.c_dtype == VT_BF16) return launch_cfg<bf16_t, bf16_t, bf16_t>(p, s);
    if (p.c_dtype == VT_F32) return launch_cfg<bf16_t, bf16_t, float>(p, s);
    return VT_ERR_UNSUPPORTED;
  }
  if (p.a_dtype == VT_F16 && p.w_dtype == VT_F16) {
    if (p.c_dtype == VT_F16) return launch_cfg<half_t, half_t, half_t>(p, s);
    if (p.c_dtype == VT_F32) return launch_cfg<half_t, half_t, float>(p, s);
    return VT_ERR_UNSUPPORTED;
  }
  if (p.a_dtype == VT_F32 && p.w_dtype == VT_BF16) {
    return p.c_dtype == VT_BF16 ? launch_cfg<float, bf16_t, bf16_t>(p, s) : launch_cfg<float, bf16_t, float>(p, s);
  }
  if (vt_gemm_f32r_eligible(p)) return vt_gemm_f32r_launch(p, s);      // exact fp32, few blocks per CU: LDS-DMA ring (vt_gemm_f32r.hip)
  if (p.a_dtype == VT_F32 && p.w_dtype == VT_F32 && p.c_dtype == VT_F32) return launch_cfg<float, float, float>(p, s);
  if (p.a_dtype == VT_F32 && p.w_dtype == VT_F32X3 && p.c_dtype == VT_F32) return launch_cfg<float, x3_t, float>(p, s);
  return VT_ERR_UNSUPPORTED;
}

// ---- profiling control (exported through include/vlatouch.h)
extern "C" int vt_prof_enable(int on) {
  g_vt_prof.on = on != 0;
  g_vt_prof.mode = on > 1 ? on : 1;
  if (on) { g_vt_prof.used = 0; g_vt_prof.flops = 0.0; g_vt_prof.bytes = 0.0; }
  return VT_OK;
}
// After the stream has been synchronised by the caller: total milliseconds, algorithmic flops and bytes, launch count.
extern "C" int vt_prof_collect(double* total_ms, double* flops, double* bytes, long* launches) {
  double ms = 0.0;
  for (int i = 0; i < g_vt_prof.used; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_vt_prof.ev[2 * i], g_vt_prof.ev[2 * i + 1]) != hipSuccess) return VT_ERR_LAUNCH;
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (flops) *flops = g_vt_prof.flops;
  if (bytes) *bytes = g_vt_prof.bytes;
  if (launches) *launches = g_vt_prof.used;
  return VT_OK;
}
