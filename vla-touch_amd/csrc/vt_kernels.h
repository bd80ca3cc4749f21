// vt_kernels.h — host launchers of the non-GEMM kernels (internal to libvlatouch_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vt_gemm.h"

enum { VT_NORM_LAYER = 0, VT_NORM_RMS_MEANSQ = 1, VT_NORM_RMS_VAR = 2 };
enum { VT_IMGNORM_AUTO = 0, VT_IMGNORM_ON = 1, VT_IMGNORM_OFF = 2 };

struct VtGnParams {
  const float* P; int nslabs; long slab_stride; long p_gs; long ldp;   // fp32 partial slabs [slab][rows][ldp]
  const float* bias; const float* gamma; const float* beta; long vec_gs;
  const float* film; long film_ld; long film_off; long film_gs;         // film[net][b][off + c] = scale, [off + C + c] = bias
  const void* residual; long ldr; long r_gs;
  void* out; long ldo; long o_gs; int out_dtype;
  int B, T, C, ngroups, nets;
  float eps;
};

struct VtAttnParams {
  const void* Q; const void* K; const void* V; void* O;
  long q_bs, q_rs, q_hs;      // element strides: batch, row(token), head
  long k_bs, k_rs, k_hs;
  long v_bs, v_rs, v_hs;
  long o_bs, o_rs;            // output [B, Nq, H*hd]
  const uint8_t* kmask; long km_bs;   // optional key mask [B, Nk] (1 = attend)
  int B, H, Nq, Nk;
  float scale;
  int dtype;
  int hd;                     // head dimension: 0 or 64 -> 64; 96
};

// cross-attention against the cached condition (bf16 only): KV = per-head tile stream over the rows b*Nk + l, see vt_attn_kvt.hip
struct VtAttnKvtParams {
  const void* Q; const void* KV; void* O;
  long q_bs, q_rs;            // Q element strides: batch, row (head h at +h*64)
  long o_bs, o_rs;
  const uint8_t* kmask;       // [B][Nk] or null
  int B, H, Nq, Nk, T;        // Nk keys per sample; T = ceil(B*Nk / 64) tiles per head
  float scale;
  int parts;                  // > 1: split every sample's key tiles over `parts` blocks (needs Nq <= 16*NW and part_ws), merged by a 2nd kernel
  float* part_ws;             // [B][H][parts][16*NW rows][66] floats
  float fixed_max;            // > 0: an upper bound of |q . k| * scale known at load time (per-head RMS-normed q and k) -> fixed-maximum softmax; 0 = online
  int dtype;                  // VT_BF16 (or 0) / VT_F16: the 16-bit type of Q, the tile stream and O
  unsigned* range_flag;       // range guard word (include/vlatouch.h): VT_RANGE_ATTN_EMPTY when a row's probabilities sum to 0 / inf (the row is written as zeros); null = none
};
// bytes of part_ws for vt_attn_kvt_launch with `parts` parts
inline size_t vt_attn_kvt_part_bytes(int B, int H, int Nq, int parts) { return parts > 1 ? (size_t)B * H * parts * ((Nq + 15) / 16 * 16 + 16) * 66 * 4 : 0; }
// position of key kk (0..63) inside a Vt tile row: within each 32-key half the keys are stored in the k order of the
// P fragment (k index g*8 + j <-> key (j>>2)*16 + g*4 + (j&3)), so an A fragment of Vt is one 16-byte chunk.  Aligned pairs
// and aligned groups of 4 keys stay contiguous.
__host__ __device__ inline int vt_kpos(int kk) { return (kk & 32) | (((kk >> 2) & 3) << 3) | (((kk >> 4) & 1) << 2) | (kk & 3); }
int vt_attn_kvt_launch(const VtAttnKvtParams& p, hipStream_t s);
void vt_attn_kvt_tune(int value);
void vt_attn16g_tune(int value);      // vt_tune(9, v): grouped-query ViT self-attention (vt_attn.hip): 0 off, 1 auto, 3 / 6 = groups per wave
// row-major K / V projections [M][ld] -> the tile stream (either source may be null)
int vt_k_retile_kv(const void* Ksrc, const void* Vsrc, long ld, void* KV, int M, int T, int H, hipStream_t s);

int vt_gemm_launch(const VtGemmParams& p, hipStream_t s);
int vt_attn_launch(const VtAttnParams& p, hipStream_t s);

// range_flag (optional): VT_RANGE_NONFINITE when a row's statistics are inf / NaN (the final norm of a ViT tower)
int vt_k_rownorm(const void* x, int xdt, long ldx, void* y, int ydt, long ldy, const float* w, const float* b, int rows, int D,
                 float eps, int mode, hipStream_t s, unsigned* range_flag = nullptr);
int vt_k_headnorm(void* x, int dt, long tok_stride, int heads, long tokens, const float* w, float eps, int mode, hipStream_t s);
int vt_k_groupnorm(const VtGnParams& p, hipStream_t s);
// out = residual + colscale * act(sum of S fp32 split-K slabs [S][M][N] + bias); residual has the output dtype
// x += sum of slabs + bias (fp32, in place), xn = rownorm(x) * w (+ b): a residual Linear and the norm that follows it (N <= 2048)
int vt_k_slab_reduce_norm(const float* slabs, int S, long slab_stride, int M, int N, const float* bias, float* x, long ldx, const float* w,
                          const float* b, float eps, int mode, void* xn, int xn_dt, long ldxn, hipStream_t s,
                          const void* pf_ptr = nullptr, size_t pf_bytes = 0);   // pf_*: the next GEMM's weights, touched by 128 extra prefetch-only blocks
// optional per-head (64 columns) RMSNorm after the bias: hn_w0 for columns [0, hn_c0), hn_w1 for [hn_c0, hn_c1) (then no act / residual use)
int vt_k_slab_reduce(const float* slabs, int S, long slab_stride, int M, int N, const float* bias, int act, const float* colscale,
                     const void* residual, long ldr, void* out, int odt, long ldo, const float* hn_w0, const float* hn_w1, int hn_c0, int hn_c1,
                     float hn_eps, int hn_mode, hipStream_t s, const void* pf_ptr = nullptr, size_t pf_bytes = 0);
int vt_k_sinusoid(const float* t, float t_host, void* out, int odt, int B, int dim, int nets, long net_stride, int rdt_style, hipStream_t s);
int vt_k_swiglu(void* h, int dt, long ld, long rows, int F, hipStream_t s, unsigned* range_flag = nullptr);       // h[r][c] = silu(h[r][c]) * h[r][F + c], c < F, in place
int vt_k_act_copy(const void* in, int idt, long ldi, void* out, int odt, long ldo, int rows, int cols, int act, hipStream_t s);
int vt_k_sde_update(float* x, const float* v, const float* sc, const float* z, long n, float dt, float gi, float gdg, float eps,
                    float noise_scale, float d, float score_eps, int backward, hipStream_t s);
int vt_k_actnorm(const float* in, float* out, const float* mins, const float* maxs, long n, int dim, float pad, int denorm, hipStream_t s);
int vt_k_pad_cols(const float* in, int cin, void* out, int odt, int cout, long rows, hipStream_t s);
int vt_k_place_cols(const void* src, int sdt, long lds_, void* out, int odt, long ldo, int off, int rows, int cols, hipStream_t s);
int vt_k_bcast_row(const float* vec, float* out, long row_stride, int B, int D, hipStream_t s);
int vt_k_imgstats(const void* img, int is_u8, long n, float pre_scale, int norm_mode, float* part, float* flags, hipStream_t s,
                  float* flags_copy = nullptr);   // flags_copy: optional second [4] output written by the same kernel
int vt_k_patchify(const void* img, int is_u8, int nhwc, int B, int res, int grid_, int kpad, const float* flags, void* out, int odt, hipStream_t s);
int vt_k_lstm_cell(const float* gi, const float* gh, float* h, float* c, int B, int H, hipStream_t s);
int vt_k_axpby3(float* x, const void* m0, const void* m1, int mdt, float a, float b0, float b1, long n, hipStream_t s);
