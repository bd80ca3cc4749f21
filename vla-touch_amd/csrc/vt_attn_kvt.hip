// vt_attn_kvt.hip — cross-attention against a CACHED condition (RDT, bf16): K row-major [B][Lk][H*64] (after k_norm) and
// V stored TRANSPOSED per head, Vt [B][H][64][Lpad] (Lpad % 64 == 0, padding zero-filled), written once per chunk by
// vt_k_transpose_v and re-read by every denoise step.  With V already transposed both operand tiles are plain 128-byte
// rows, so they go HBM -> LDS by DMA (global_load_lds_dwordx4, source-side XOR swizzle) into a 2-stage ring while the
// previous tile's MFMAs run: the kernel streams the 1.15 GB of RDT-1B image K/V per call at HBM rate instead of
// staging through registers with scalar LDS transposes (vt_attn.hip, still used for self-attention).
// Block = NW waves = 16*NW query rows of one (batch, head); fragment conventions as vt_attn.hip.
#include "vt_common.h"
#include "vt_kernels.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr int KT = 64;                 // keys per tile
constexpr int STAGE = 2 * KT * 128;    // K tile (64 rows x 128 B) + Vt tile (64 d-rows x 128 B) = 16 KiB

__global__ __launch_bounds__(512) void attn_kvt_kernel(const VtAttnKvtParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * (nw * 16) + wave * 16 + l15;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.Q) + (long)b * p.q_bs + (long)h * 64;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.K) + (long)b * p.k_bs + (long)h * 64;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(p.VT) + ((long)b * p.H + h) * 64 * p.Lpad;
  const uint8_t* km = p.kmask ? p.kmask + (long)b * p.Nk : nullptr;

  Frag<bf16_t> qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    qf[ks].v = q < p.Nq ? *reinterpret_cast<const short8_t*>(Q + (long)q * p.q_rs + ks * 32 + g * 8) : (short8_t){0, 0, 0, 0, 0, 0, 0, 0};

  // DMA plan: 16 wave-instructions per tile (8 for K rows, 8 for Vt rows), instruction i handled by wave i % nw.
  // lane -> (row = i8*8 + lane/8, chunk position = lane%8); it fetches the chunk whose swizzled position is its own.
  const int r_in = lane >> 3, pch = lane & 7;
  auto stage = [&](int buf, int tile) {
    char* base = smem + buf * STAGE;
    const int key0 = tile * KT;
    for (int i = wave; i < 16; i += nw) {
      const int i8 = i & 7;
      const int r = i8 * 8 + r_in;
      const int c = pch ^ ((r >> 1) & 7);
      if (i < 8) {
        const int key = min(key0 + r, p.Nk - 1);                     // clamped rows are masked in the softmax
        __builtin_amdgcn_global_load_lds((glb_void*)(K + (long)key * p.k_rs + c * 8), (lds_void*)(base + i8 * 1024), 16, 0, 0);
      } else {
        __builtin_amdgcn_global_load_lds((glb_void*)(VT + (long)r * p.Lpad + key0 + c * 8), (lds_void*)(base + KT * 128 + i8 * 1024), 16, 0, 0);
      }
    }
  };

  float4_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (p.Nk + KT - 1) / KT;
  stage(0, 0);
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int cur = tile & 1;
    if (tile + 1 < ntiles) stage(cur ^ 1, tile + 1);
    const char* Ks = smem + cur * STAGE;
    const char* Vs = Ks + KT * 128;
    const int key0 = tile * KT;

    float4_t sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        Frag<bf16_t> kf;
        lds_frag(kf, Ks, kt * 16 + l15, ks * 4 + g);
        mma16(sacc[kt], kf, qf[ks]);
      }
    }
    float sv[16];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kidx = key0 + kt * 16 + g * 4 + r;
        bool ok = kidx < p.Nk;
        if (ok && km) ok = km[kidx] != 0;
        const float s = ok ? sacc[kt][r] * p.scale : -INFINITY;
        sv[kt * 4 + r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_use);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { sv[i] = __expf(sv[i] - m_use); psum += sv[i]; }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      Frag<bf16_t> pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf.v[j] = (short)f2bf(sv[(kb * 2 + (j >> 2)) * 4 + (j & 3)]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // Vt row d = dt*16 + l15; keys kb*32 + g*4 .. +3 (8 B) and kb*32 + 16 + g*4 .. +3: chunk = key/8, half = g&1
        const int row = dt * 16 + l15;
        const int sw = (row >> 1) & 7;
        const char* rp = Vs + row * 128 + (g & 1) * 8;
        const uint2 lo = *reinterpret_cast<const uint2*>(rp + (((kb * 4 + (g >> 1)) ^ sw) << 4));
        const uint2 hi = *reinterpret_cast<const uint2*>(rp + (((kb * 4 + 2 + (g >> 1)) ^ sw) << 4));
        Frag<bf16_t> vf;
        vf.v[0] = (short)(lo.x & 0xffff); vf.v[1] = (short)(lo.x >> 16); vf.v[2] = (short)(lo.y & 0xffff); vf.v[3] = (short)(lo.y >> 16);
        vf.v[4] = (short)(hi.x & 0xffff); vf.v[5] = (short)(hi.x >> 16); vf.v[6] = (short)(hi.y & 0xffff); vf.v[7] = (short)(hi.y >> 16);
        mma16(o[dt], vf, pf);
      }
    }
    __syncthreads();      // next stage landed (the barrier drains the DMA) and this stage is free again
  }
  float l = l_run;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = 1.0f / l;
  if (q < p.Nq) {
    bf16_t* O = reinterpret_cast<bf16_t*>(p.O) + (long)b * p.o_bs + (long)q * p.o_rs + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 t;
      t.x = (uint32_t)f2bf(o[dt][0] * inv) | ((uint32_t)f2bf(o[dt][1] * inv) << 16);
      t.y = (uint32_t)f2bf(o[dt][2] * inv) | ((uint32_t)f2bf(o[dt][3] * inv) << 16);
      *reinterpret_cast<uint2*>(O + dt * 16 + g * 4) = t;
    }
  }
}

// V [B][L][ld] (head h at columns h*64..) -> Vt [B][H][64][Lpad], zero padded for l >= L.  One block per (64-token tile, h, b).
__global__ __launch_bounds__(256) void transpose_v_kernel(const bf16_t* __restrict__ V, long ld, bf16_t* __restrict__ VT, int L, int Lpad, int H) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.z, h = blockIdx.y, l0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const bf16_t* src = V + (long)b * L * ld + (long)h * 64;
  for (int e = tid; e < 64 * 8; e += 256) {            // 64 tokens x 8 chunks of 8 d
    const int t = e >> 3, c = e & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (l0 + t < L) v = *reinterpret_cast<const uint4*>(src + (long)(l0 + t) * ld + c * 8);
    const bf16_t* ve = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[c * 8 + j][t] = ve[j];
  }
  __syncthreads();
  bf16_t* dst = VT + ((long)b * H + h) * 64 * Lpad + l0;
  for (int e = tid; e < 64 * 32; e += 256) {           // 64 d rows x 32 pairs of tokens
    const int d = e >> 5, t2 = (e & 31) * 2;
    const uint32_t v = (uint32_t)tile[d][t2] | ((uint32_t)tile[d][t2 + 1] << 16);
    *reinterpret_cast<uint32_t*>(dst + (long)d * Lpad + t2) = v;
  }
}

}  // namespace

int vt_attn_kvt_launch(const VtAttnKvtParams& p, hipStream_t s) {
  if (p.B <= 0 || p.H <= 0 || p.Nq <= 0 || p.Nk <= 0 || p.Lpad % 64 || p.Lpad < p.Nk || p.q_rs % 8 || p.k_rs % 8) return VT_ERR_ARG;
  int nw = 4, best = 1 << 30;
  for (int w = 4; w <= 8; ++w) {
    const int rows = w * 16, padded = (p.Nq + rows - 1) / rows * rows;
    if (padded < best) { best = padded; nw = w; }
  }
  dim3 grid((p.Nq + nw * 16 - 1) / (nw * 16), p.H, p.B);
  hipLaunchKernelGGL(attn_kvt_kernel, grid, dim3(64 * nw), 0, s, p);
  return vt_check_launch();
}

int vt_k_transpose_v(const void* V, long ld, void* VT, int B, int L, int Lpad, int H, hipStream_t s) {
  if (Lpad % 64 || Lpad < L || ld % 8) return VT_ERR_ARG;
  hipLaunchKernelGGL(transpose_v_kernel, dim3(Lpad / 64, H, B), dim3(256), 0, s, (const bf16_t*)V, ld, (bf16_t*)VT, L, Lpad, H);
  return vt_check_launch();
}
