// vt_unet_int.h — handle layout of the conditional 1-D U-Nets, shared by the launch-per-op driver (vt_unet.hip) and the fused driver
// (vt_unet_fused.hip).  Internal to libvlatouch_hip.so.
#pragma once
#include <stdint.h>
#include "../../include/vlatouch.h"

// a convolution of the fused path: weights in the fragment-ordered hi / lo stream of vt_uconv.h
struct FConv { const uint16_t* wp; long w_gs, w_ps; const uint16_t* wr; long wr_gs; int nc32, ntaps, has_res, N; };

struct ResBlk {
  int cin, cin_pad, cout;
  const void *c0_w, *c1_w, *res_w;
  const float *c0_b, *g0, *be0, *c1_b, *g1, *be1, *res_b;
  long film_off;
};

struct vt_unet_s {
  vt_unet_desc d;
  int nrb;
  ResBlk rb[32];
  const void *step_w1, *step_w2, *film_w;
  const float *step_b1, *step_b2, *film_b;
  const void* down_w[8]; const float* down_b[8];
  const void *up_we[8], *up_wo[8]; const float* up_b[8];
  const void *fc_w, *out_w; const float *fc_b, *fg, *fbe, *out_b;
  long F;
  int cmax;
  // fused path (vt_uconv.hip): set by vt_unet_fused_pack
  bool fused;
  FConv f_c0[32], f_c1[32], f_down[8], f_up[8], f_fc;
};

void vt_unet_fused_tune(int on);   // vt_tune knob 7

// fused driver (vt_unet_fused.hip)
size_t vt_unet_fused_workspace_bytes(const vt_unet_s* h, int B, int T, int n_steps);
bool vt_unet_fused_ok(const vt_unet_s* h, int B, int T, int n_steps);
void vt_unet_fused_init_meta(vt_unet_s* h);       // called by vt_unet_create
// n_steps evaluations of the nets at the scalar times ts[k] on the evolving state x: with `sde` the Euler-Maruyama update of step k is applied
// by the last kernel (coefficient arrays indexed by k); without, n_steps must be 1 and vs_out receives the raw net outputs
struct VtSdeCoef { float dt, gi, gdg, eps_t, noise_scale, d, score_eps; int backward; };
int vt_unet_fused_run(const vt_unet_s* h, float* x, const float* cond, const float* ts, const VtSdeCoef* coef, int n_steps, const float* noise,
                      float* traj, float* vs_out, int B, int T, void* ws, hipStream_t s);
