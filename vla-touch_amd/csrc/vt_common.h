// vt_common.h — shared device helpers for the gfx950 (CDNA4) kernels of libvlatouch_hip.so.
// Wave = 64 lanes.  MFMA fragment convention used everywhere in this library ("8-consecutive-k"):
//   a lane (row = lane & 15, g = lane >> 4) holds the 8 elements k = g*8 .. g*8+7 of a 32-wide k-step
//   for BOTH operands; C/D: lane holds D[(lane>>4)*4 + r][lane & 15], r = 0..3.
//   bf16 : one v_mfma_f32_16x16x32_bf16 per k-step.
//   f32  : eight v_mfma_f32_16x16x4_f32 (element j of every lane = one 4-deep slice); any consistent
//          k -> (slice, lane-group) assignment sums the same products, so both types share addressing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef uint16_t bf16_t;  // raw bits
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
struct half_t { uint16_t b; };   // raw IEEE binary16 bits, a distinct type so templates can tell it from bf16_t

#define VT_OK 0
#define VT_ERR_ARG (-22)
#define VT_ERR_LAUNCH (-5)
#define VT_ERR_UNSUPPORTED (-95)

// F32X3: fp32 storage, split-bf16 3-MFMA compute (GEMM weights only).  F16: IEEE half storage + v_mfma_f32_16x16x32_f16
// (same rate as bf16, 3 more mantissa bits; used where activations are range-bounded, i.e. the DINOv2 encoders).
enum { VT_F32 = 0, VT_BF16 = 1, VT_F32X3 = 2, VT_F16 = 3 };
enum { VT_ACT_NONE = 0, VT_ACT_GELU_ERF = 1, VT_ACT_GELU_TANH = 2, VT_ACT_SILU = 3, VT_ACT_MISH = 4,
       VT_ACT_SWIGLU = 5 };    // not an element-wise epilogue: a ViT FFN form (vt_dino_desc.act): fc1 -> [x1 | x2], silu(x1) * x2, fc2 (HF Dinov2SwiGLUFFN)

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
typedef __attribute__((ext_vector_type(2))) float float2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// float -> bf16, round-to-nearest-even, NaN preserved: v_cvt_pk_bf16_f32 (one instruction per PAIR)
__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float h2f(half_t h) { return (float)__builtin_bit_cast(_Float16, h.b); }
__device__ __forceinline__ half_t f2h(float f) { half_t r; r.b = __builtin_bit_cast(uint16_t, (_Float16)f); return r; }   // v_cvt_f16_f32: RNE, saturates to inf

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
  __device__ static __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};
template <> struct Elem<half_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float to_f(half_t v) { return h2f(v); }
  __device__ static __forceinline__ half_t from_f(float v) { return f2h(v); }
};

// 8-element fragment of compute type T
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { short8_t v; };
template <> struct Frag<half_t> { short8_t v; };
template <> struct Frag<float> { float v[8]; };

__device__ __forceinline__ void mma16(float4_t& acc, const Frag<bf16_t>& a, const Frag<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(float4_t& acc, const Frag<half_t>& a, const Frag<half_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a.v), __builtin_bit_cast(f16x8_t, b.v), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(float4_t& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// read an 8-element fragment from LDS rows of 8 16-byte chunks (128 B), chunk XOR-swizzled by row.
// `k8` = index of the 8-element group inside the row (0 .. 128/ (8*sizeof(T)) - 1).
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

__device__ __forceinline__ void lds_frag(Frag<bf16_t>& f, const char* tile, int row, int k8) {
  f.v = *reinterpret_cast<const short8_t*>(tile + row * 128 + swz(row, k8) * 16);
}
__device__ __forceinline__ void lds_frag(Frag<half_t>& f, const char* tile, int row, int k8) {
  f.v = *reinterpret_cast<const short8_t*>(tile + row * 128 + swz(row, k8) * 16);
}
__device__ __forceinline__ void lds_frag(Frag<float>& f, const char* tile, int row, int k8) {
  const float4_t lo = *reinterpret_cast<const float4_t*>(tile + row * 128 + swz(row, 2 * k8) * 16);
  const float4_t hi = *reinterpret_cast<const float4_t*>(tile + row * 128 + swz(row, 2 * k8 + 1) * 16);
  f.v[0] = lo[0]; f.v[1] = lo[1]; f.v[2] = lo[2]; f.v[3] = lo[3];
  f.v[4] = hi[0]; f.v[5] = hi[1]; f.v[6] = hi[2]; f.v[7] = hi[3];
}

// exp(x) on the hardware exp2 unit (v_exp_f32, ~1 ulp): the activations below are evaluated per output element of GEMM / GroupNorm
// epilogues, where libm's tanhf / log1pf / expf (30-60 instructions each) cost more than the GEMM's k-loop saves
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. about one fp32 ulp of erf near 1; exact GELU differs from it by
// <= 0.5 |x| 1.5e-7) on the exp2 / rcp units: ~14 instructions where libm's erff is ~60 with branches.  The fc1 epilogue of a ViT applies
// it to M x 4D elements per layer: with erff the DINOv2-base fc1 launch (16448 x 3072 x 768) took 185 us against 99 us without activation.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float p = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  return copysignf(1.0f - p * fast_exp(-ax * ax), x);
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case VT_ACT_GELU_ERF: return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f));
    case VT_ACT_GELU_TANH: {
      // 0.5 x (1 + tanh(u)) == x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)
      const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);
      return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-u2));
    }
    case VT_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + fast_exp(-x));
    case VT_ACT_MISH: {
      // x tanh(log(1 + e^x)) == x w / (w + 2), w = e^x (e^x + 2)   (tanh(log s) = (s^2 - 1) / (s^2 + 1), s = 1 + e^x)
      if (x > 20.0f) return x;
      const float n = fast_exp(x), w = n * (n + 2.0f);
      return x * w * __builtin_amdgcn_rcpf(w + 2.0f);
    }
    default: return x;
  }
}

// sum / max over the 64 lanes, result in every lane.  Round 5: inside a row of 16 on DPP lane permutes, across the four rows on v_permlane16_swap /
// v_permlane32_swap — all VALU — instead of six __shfl_xor = ds_bpermute round trips through the LDS crossbar (-DVLATOUCH_WAVE_SHFL keeps the butterfly for A/B).
__device__ __forceinline__ float rows4_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
  const unsigned x = __builtin_bit_cast(unsigned, w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}
__device__ __forceinline__ float rows4_max(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float w = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
  const unsigned x = __builtin_bit_cast(unsigned, w);
  const auto b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over each aligned group of 16 lanes, result in every lane, on DPP lane permutes only (no LDS crossbar round trips as
// __shfl_xor would take): pairs inside quads, quads inside 8s (row_half_mirror pairs quad 0 <-> quad 1), 8s inside the row of 16
// (row_mirror); for a sum any pairing that covers the group works.
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0xB1>{}));
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x4E>{}));
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x141>{}));
  v = fmaxf(v, dpp(v, std::integral_constant<int, 0x140>{}));
  return v;
}
#ifdef VLATOUCH_WAVE_SHFL
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_shfl(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#else
__device__ __forceinline__ float wave_sum(float v) { return rows4_sum(row16_sum(v)); }
__device__ __forceinline__ float wave_max(float v) { return rows4_max(row16_max(v)); }
#endif

// generic typed load/store as float
template <typename T> __device__ __forceinline__ float ldf(const T* p, size_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p, size_t i) { return bf2f(p[i]); }
template <> __device__ __forceinline__ float ldf<half_t>(const half_t* p, size_t i) { return h2f(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, size_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, size_t i, float v) { p[i] = f2bf(v); }
template <> __device__ __forceinline__ void stf<half_t>(half_t* p, size_t i, float v) { p[i] = f2h(v); }

// (bit values: include/vlatouch.h)
#ifndef VT_RANGE_XN_SAT
#define VT_RANGE_XN_SAT 1u
#define VT_RANGE_NONFINITE 2u
#define VT_RANGE_GATE_SAT 4u
#define VT_RANGE_ATTN_EMPTY 8u
#endif
// sticky range-guard word of an engine (include/vlatouch.h: vt_rdt_set_range_flag): kernels OR bits in where a value left the 16-bit type's range; rare, relaxed, no return value
__device__ __forceinline__ void vt_range_note(unsigned* flag, unsigned bit) {
  if (flag) __hip_atomic_fetch_or(flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool vt_nonfinite(float v) { return !(fabsf(v) <= 3.4028234664e38f); }      // inf or NaN

static inline int vt_check_launch() { return hipGetLastError() == hipSuccess ? VT_OK : VT_ERR_LAUNCH; }
