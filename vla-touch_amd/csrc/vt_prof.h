// vt_prof.h — live per-launch timing of the LDS-DMA GEMM kernels (bench.py's roofline leg): while enabled, every launch of
// the selected kernel class is bracketed by HIP events recorded on its launch stream.
//   vt_prof_enable(0) off; (1) both classes; (2) only gemm_pp256_kernel (vt_gemm_pp.hip); (3) only gemm_glds_kernel (vt_gemm_fast.hip);
//   (4) only the cached cross-attention (vt_attn_kvt.hip); (5) only the register-staged gemm_kernel (vt_gemm.hip);
//   (6) only the fused U-Net convolution (vt_uconv.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "vt_common.h"
#include "vt_gemm.h"

struct VtProfState {
  bool on = false;
  int mode = 1;
  static constexpr int MAXEV = 8192;
  hipEvent_t ev[2 * MAXEV];
  int created = 0, used = 0;
  double flops = 0.0, bytes = 0.0;
};
extern VtProfState g_vt_prof;

struct VtProfScope {
  bool active; hipStream_t s; int idx;
  VtProfScope(int cls, const VtGemmParams& p, hipStream_t st) : active(g_vt_prof.on && (g_vt_prof.mode == 1 || g_vt_prof.mode == cls) && g_vt_prof.used < VtProfState::MAXEV), s(st), idx(0) {
    if (!active) return;
    idx = g_vt_prof.used++;
    while (g_vt_prof.created < 2 * (idx + 1)) (void)hipEventCreate(&g_vt_prof.ev[g_vt_prof.created++]);
    g_vt_prof.flops += 2.0 * p.M * (double)p.N * p.K * p.groups;
    g_vt_prof.bytes += ((double)p.M * p.K + (double)p.N * p.K) * 2.0 * p.groups + (double)p.M * p.N * p.groups * (p.c_dtype == VT_F32 ? 4.0 : 2.0);
    (void)hipEventRecord(g_vt_prof.ev[2 * idx], s);
  }
  // a kernel that is not a GEMM: the caller states its algorithmic flops / bytes (class 4 = cached cross-attention, HBM-bound)
  VtProfScope(int cls, double flops, double bytes, hipStream_t st) : active(g_vt_prof.on && g_vt_prof.mode == cls && g_vt_prof.used < VtProfState::MAXEV), s(st), idx(0) {
    if (!active) return;
    idx = g_vt_prof.used++;
    while (g_vt_prof.created < 2 * (idx + 1)) (void)hipEventCreate(&g_vt_prof.ev[g_vt_prof.created++]);
    g_vt_prof.flops += flops; g_vt_prof.bytes += bytes;
    (void)hipEventRecord(g_vt_prof.ev[2 * idx], s);
  }
  ~VtProfScope() { if (active) (void)hipEventRecord(g_vt_prof.ev[2 * idx + 1], s); }
};

bool vt_gemm_fast_eligible(const VtGemmParams& p);
bool vt_gemm_can_fuse_headnorm(const VtGemmParams& p);
bool vt_gemm_pp_eligible(const VtGemmParams& p);          // vt_gemm_pp.hip: 256-square ping-pong tile
int vt_gemm_pp_launch(const VtGemmParams& p, hipStream_t s);
bool vt_gemm_pp_shape(const VtGemmParams& p);             // what gemm_pp256d_kernel itself is eligible for
bool vt_gemm_pt_extra_shape(const VtGemmParams& p);       // shapes only the persistent kernel takes (one round of 160 .. 256 tiles at K >= 512)
bool vt_gemm_pt_eligible(const VtGemmParams& p);          // vt_gemm_pt.hip: the same tile, persistent, epilogue on registers inside the main loop
int vt_gemm_pt_launch(const VtGemmParams& p, hipStream_t s);
void vt_gemm_pt_tune(int value);                           // 1 = on (default; env VLATOUCH_PT), 0 = off (gemm_pp256d_kernel takes those launches)
bool vt_gemm_ppk_eligible(const VtGemmParams& p);         // vt_gemm_ppk.hip: 160 x 128 tile, in-block split-K ping-pong, one round
int vt_gemm_ppk_launch(const VtGemmParams& p, hipStream_t s);
bool vt_gemm_pw_eligible(const VtGemmParams& p);          // vt_gemm_pw.hip: 160 x 128 tile, fragment-packed weights streamed global -> VGPR
int vt_gemm_pw_launch(const VtGemmParams& p, hipStream_t s);
bool vt_gemm_pws_eligible(const VtGemmParams& p);         // vt_gemm_pws.hip: M <= 512, 96 x 64 tiles, split-K with an in-kernel ticket reduction
int vt_gemm_pws_launch(const VtGemmParams& p, hipStream_t s);
void vt_gemm_pws_tune(int knob, int value);
int vt_gemm_fast_launch(const VtGemmParams& p, hipStream_t s);
bool vt_gemm_f32r_eligible(const VtGemmParams& p);        // vt_gemm_f32r.hip: exact fp32, 64 x 64 x 32 tiles through an LDS-DMA ring
int vt_gemm_f32r_launch(const VtGemmParams& p, hipStream_t s);
