// vt_marker.hip — GelSight marker tracker on the device: frames -> marker centroids -> displacements / force estimate m_t
// (replaces residual_controller/tactile/marker/marker_tracker.py:81-114 init_standard, :154-183 detect_markers,
//  :308-341 match_and_compute_displacement, :343-373 estimate_force; the cv2 primitives behind them are restated from
//  OpenCV's published algorithms — oracle/marker.py carries the same statements and the citations).
//
// Byte / integer work, HBM-bound (77 KB per 320x240 frame in, a few hundred bytes out): no MFMA anywhere.  Batched over
// frames (an episode's GelSight stream is labelled in one call):
//   1. marker_binary_kernel   one block per 32x32 output tile of one frame: the tile plus a 9-pixel halo is staged ONCE in LDS
//        (coalesced byte loads, BGR->gray fixed point on the fly) and the whole chain runs out of LDS:
//        gray -> 5x5 binomial blur (integer, REFLECT_101) -> 11x11 Gaussian mean (separable fp32, REPLICATE, products and sums
//        rounded separately like the CPU statement) -> threshold (blur - rint(mean) <= -2) -> 3x3 erode -> 3x3 dilate -> 1 byte/pixel.
//   2. marker_label_kernels   8-connected components by union-find on the label image (label = raster index of the
//        component's first pixel = where OpenCV's border following starts).
//   3. marker_trace_kernel    one lane per component root: Moore border following of the OUTER border, Green's-theorem
//        polygon sums in int64 (exact), area filter, centroid = int(m10/m00), int(m01/m00) in fp64.
//   4. marker_order_kernel    candidates -> cv2.findContours order (last found first = descending root index).
//   5. marker_disp_kernel     nearest baseline marker (ties -> lowest index), displacement, mean -> |F|, direction (fp64).
#include <math.h>
#include <string.h>
#include "vt_common.h"
#include "vt_host.h"
#include "../../include/vlatouch.h"

namespace {

constexpr int TS = 32;                    // output tile
constexpr int HALO = 9;                   // 2 (open) + 5 (mean) + 2 (blur)
constexpr int GW = TS + 2 * HALO;         // gray tile 50
constexpr int BW = TS + 2 * 7;            // blurred tile 46 (halo 7)
constexpr int MW = TS + 2 * 2;            // thresholded tile 36 (halo 2)
constexpr int EW = TS + 2;                // eroded tile 34

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);        // degenerate sizes: clamp
}
__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

struct MeanKernel { float k[11]; };

// gray value of pixel (y, x) of a frame: BGR -> gray fixed point, or the single channel.  `channels` carries the coefficient set in bit 8:
// clear = OpenCV >= 3.4.2 / 4.x RGB2Gray<uchar> (15-bit: BY15 3735, GY15 19235, RY15 9798, CV_DESCALE(., 15)) — what an unpinned
// `opencv-python` resolves to today —, set = OpenCV <= 3.4.1 (14-bit table: B2Y 1868, G2Y 9617, R2Y 4899, + 8192 >> 14).  The two differ
// by one grey level on ~0.3 % of random colour pixels, never on grey ones.
__device__ __forceinline__ int gray_at(const uint8_t* F, int channels, int W, int y, int x) {
  const int ch = channels & 0xff;
  const uint8_t* px = F + ((size_t)y * W + x) * ch;
  if (ch != 3) return px[0];
  return (channels & 0x100) ? ((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14) : ((px[0] * 3735 + px[1] * 19235 + px[2] * 9798 + 16384) >> 15);
}

// 'HSR' sensor (init_HSR): histogram of the INVERTED gray image, one block per (frame, slab of rows)
__global__ __launch_bounds__(256) void marker_hist_kernel(const uint8_t* __restrict__ frames, int channels, int H, int W, int* __restrict__ hist) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint8_t* F = frames + (size_t)blockIdx.y * H * W * (channels & 0xff);
  const int n = H * W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) atomicAdd(&h[255 - gray_at(F, channels, W, i / W, i % W)], 1);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[blockIdx.y * 256 + threadIdx.x], h[threadIdx.x]);      // integer adds: order-independent
}
// cv::equalizeHist look-up table per frame (256 entries; a constant image maps to itself)
__global__ void marker_lut_kernel(const int* __restrict__ hist, int total, uint8_t* __restrict__ lut) {
  if (threadIdx.x != 0) return;
  const int* h = hist + blockIdx.x * 256;
  uint8_t* l = lut + blockIdx.x * 256;
  int i0 = 0;
  while (i0 < 255 && h[i0] == 0) ++i0;
  for (int i = 0; i < 256; ++i) l[i] = 0;
  if (h[i0] == total) { l[i0] = (uint8_t)i0; return; }
  const float scale = 255.0f / (float)(total - h[i0]);
  int sum = 0;
  for (int i = i0 + 1; i < 256; ++i) {
    sum += h[i];
    int v = (int)rintf(__fmul_rn((float)sum, scale));
    l[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

// HSR = false: init_standard (adaptive Gaussian threshold of the blurred gray image); HSR = true: init_HSR (gray -> 255 - gray ->
// equalisation LUT -> blur -> fixed threshold > 50).  Same tile / halo plan for both.
template <bool HSR>
__global__ __launch_bounds__(256) void marker_binary_kernel(const uint8_t* __restrict__ frames, int channels, int H, int W, MeanKernel mk,
                                                            const uint8_t* __restrict__ lut, uint8_t* __restrict__ binary) {
  __shared__ uint8_t gray[GW][GW + 2];
  __shared__ uint8_t blur[BW][BW + 2];
  __shared__ float hmean[BW][MW + 1];
  __shared__ uint8_t thr[MW][MW + 4];
  __shared__ uint8_t ero[EW][EW + 2];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const uint8_t* F = frames + (size_t)blockIdx.z * H * W * (channels & 0xff);
  // ---- gray tile: tile position (r, c) <-> image (y0 - 9 + r, x0 - 9 + c), out-of-image positions hold the REFLECT_101 pixel
  for (int e = tid; e < GW * GW; e += 256) {
    const int r = e / GW, c = e - r * GW;
    const int y = reflect101(y0 - HALO + r, H), x = reflect101(x0 - HALO + c, W);
    const int gv = gray_at(F, channels, W, y, x);
    gray[r][c] = HSR ? lut[blockIdx.z * 256 + 255 - gv] : (uint8_t)gv;
  }
  __syncthreads();
  // ---- blurred tile (halo 7): position (r, c) <-> image (y0 - 7 + r, x0 - 7 + c); outside the image the adaptive mean sees the
  // REPLICATEd blurred image, i.e. the blur evaluated at the CLAMPED coordinate (whose own neighbourhood is in the gray tile)
  for (int e = tid; e < BW * BW; e += 256) {
    const int r = e / BW, c = e - r * BW;
    const int yc = clampi(y0 - 7 + r, H), xc = clampi(x0 - 7 + c, W);
    const int gr = yc - (y0 - HALO), gc = xc - (x0 - HALO);
    int acc = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
      const int wy = dy == 0 ? 6 : (dy == -1 || dy == 1 ? 4 : 1);
      const uint8_t* g = &gray[gr + dy][gc];
      acc += wy * (g[-2] + 4 * g[-1] + 6 * g[0] + 4 * g[1] + g[2]);
    }
    blur[r][c] = (uint8_t)((acc + 128) >> 8);
  }
  __syncthreads();
  // ---- adaptive mean, horizontal pass: hmean (r, c) <-> image row y0 - 7 + r, column x0 - 2 + c
  if (!HSR)
  for (int e = tid; e < BW * MW; e += 256) {
    const int r = e / MW, c = e - r * MW;
    float h = 0.f;
#pragma unroll
    for (int i = 0; i < 11; ++i) h = __fadd_rn(h, __fmul_rn(mk.k[i], (float)blur[r][c + i]));
    hmean[r][c] = h;
  }
  __syncthreads();
  // ---- vertical pass + threshold: thr (r, c) <-> image (y0 - 2 + r, x0 - 2 + c); outside the image: erode's border (never wins)
  for (int e = tid; e < MW * MW; e += 256) {
    const int r = e / MW, c = e - r * MW;
    const int y = y0 - 2 + r, x = x0 - 2 + c;
    const bool inside = y >= 0 && y < H && x >= 0 && x < W;
    if (HSR) {
      thr[r][c] = !inside ? 2 : (blur[r + 5][c + 5] > 50 ? 1 : 0);
      continue;
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 11; ++i) v = __fadd_rn(v, __fmul_rn(mk.k[i], hmean[r + i][c]));
    int mean = (int)rintf(v);
    mean = mean < 0 ? 0 : (mean > 255 ? 255 : mean);
    thr[r][c] = !inside ? 2 : (((int)blur[r + 5][c + 5] - mean <= -2) ? 1 : 0);       // 2 = outside the image
  }
  __syncthreads();
  // ---- erode (min over 3x3; outside = +inf): ero (r, c) <-> image (y0 - 1 + r, x0 - 1 + c); outside the image: dilate's border (0)
  for (int e = tid; e < EW * EW; e += 256) {
    const int r = e / EW, c = e - r * EW;
    const int y = y0 - 1 + r, x = x0 - 1 + c;
    uint8_t m = 1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) { const uint8_t t = thr[r + dy][c + dx]; if (t == 0) m = 0; }
    ero[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? m : 0;
  }
  __syncthreads();
  // ---- dilate (max over 3x3) -> binary
  uint8_t* Bz = binary + (size_t)blockIdx.z * H * W;
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e / TS, c = e - r * TS;
    const int y = y0 + r, x = x0 + c;
    if (y >= H || x >= W) continue;
    uint8_t m = 0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) m |= ero[r + dy][c + dx];
    Bz[(size_t)y * W + x] = m;
  }
}

// ---- 8-connected components: union-find on the label image (Playne-Hawick style; root = smallest raster index)
__device__ __forceinline__ int uf_find(const int* L, int i) {
  int p = L[i];
  while (p != i) { i = p; p = L[i]; }
  return i;
}
__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }        // a > b: hang a under b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}
__global__ void marker_nonzero_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] ? 1 : 0;
}
// labels are frame-local raster indices (each frame has its own [H*W] slab)
__global__ void marker_label_merge_kernel(const uint8_t* __restrict__ binary, int* __restrict__ Lall, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const uint8_t* B = binary + (size_t)blockIdx.y * H * W;
  int* L = Lall + (size_t)blockIdx.y * H * W;
  if (!B[i]) return;
  const int y = i / W, x = i - y * W;
  if (x > 0 && B[i - 1]) uf_union(L, i, i - 1);
  if (y > 0) {
    if (B[i - W]) uf_union(L, i, i - W);
    if (x > 0 && B[i - W - 1]) uf_union(L, i, i - W - 1);
    if (x + 1 < W && B[i - W + 1]) uf_union(L, i, i - W + 1);
  }
}
__global__ void marker_label_local_init_kernel(const uint8_t* __restrict__ binary, int* __restrict__ Lall, int HW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  Lall[(size_t)blockIdx.y * HW + i] = binary[(size_t)blockIdx.y * HW + i] ? i : -1;
}

struct Cand { int root, cx, cy, pad; };

// ---- one lane per component root: outer border by Moore neighbour tracing from the raster-first pixel, polygon sums, filter
__global__ void marker_trace_kernel(const uint8_t* __restrict__ binary, const int* __restrict__ Lall, int H, int W, double min_area, double max_area,
                                    Cand* __restrict__ cand, int* __restrict__ ncand, int max_cand) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * W) return;
  const int f = blockIdx.y;
  const uint8_t* B = binary + (size_t)f * H * W;
  if (Lall[(size_t)f * H * W + i] != i) return;            // not a root (or background)
  const int NBX[8] = {1, 1, 0, -1, -1, -1, 0, 1}, NBY[8] = {0, 1, 1, 1, 0, -1, -1, -1};   // clockwise from East (y down)
  const int x0 = i % W, y0 = i / W;
  auto fg = [&](int x, int y) { return x >= 0 && x < W && y >= 0 && y < H && B[(size_t)y * W + x]; };
  long long a00 = 0, a10 = 0, a01 = 0;
  int cx = x0, cy = y0, d = 4, first_dir = -1;
  int px = x0, py = y0;                                     // previous polygon vertex
  long steps = 0;
  const long max_steps = 4L * H * W;
  bool moved = false;
  while (steps++ < max_steps) {
    int k = -1, nx = 0, ny = 0;
    for (int s = 0; s < 8; ++s) {
      const int kk = (d + s) & 7;
      nx = cx + NBX[kk]; ny = cy + NBY[kk];
      if (fg(nx, ny)) { k = kk; break; }
    }
    if (k < 0) break;                                       // isolated pixel
    if (cx == x0 && cy == y0) {
      if (first_dir < 0) first_dir = k;
      else if (k == first_dir && moved) break;              // back at the start, leaving as the first time
    }
    // polygon edge (px,py) -> (nx,ny): vertex list is start, p1, p2, ..., closing edge last -> start is added below
    cx = nx; cy = ny; moved = true;
    const long long dxy = (long long)px * cy - (long long)cx * py;
    a00 += dxy; a10 += dxy * (px + cx); a01 += dxy * (py + cy);
    px = cx; py = cy;
    d = (k + 5) & 7;
  }
  // The loop adds the edge into every visited vertex including the final return to the start (the repeated start vertex is the
  // closing edge of the polygon), so the sums are complete.
  if (a00 == 0) return;
  const double sgn = a00 > 0 ? 1.0 : -1.0;
  const double m00 = sgn * (double)a00 * 0.5, m10 = sgn * (double)a10 / 6.0, m01 = sgn * (double)a01 / 6.0;
  if (!(min_area < m00 && m00 < max_area)) return;
  const int slot = atomicAdd(&ncand[f], 1);
  if (slot < max_cand) cand[(size_t)f * max_cand + slot] = Cand{i, (int)(m10 / m00), (int)(m01 / m00), 0};
}

// candidates -> markers in cv2.findContours order (descending root index); one block per frame
__global__ __launch_bounds__(256) void marker_order_kernel(const Cand* __restrict__ cand, int* __restrict__ ncand, int max_cand, int* __restrict__ markers,
                                                            int* __restrict__ counts, int max_markers) {
  const int f = blockIdx.x;
  const int n = min(ncand[f], max_cand);
  const Cand* C = cand + (size_t)f * max_cand;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const int root = C[e].root;
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += C[j].root > root;
    if (rank < max_markers) { markers[((size_t)f * max_markers + rank) * 2] = C[e].cx; markers[((size_t)f * max_markers + rank) * 2 + 1] = C[e].cy; }
  }
  if (threadIdx.x == 0) counts[f] = n;                   // may exceed max_markers: the host sees the overflow
}

// nearest baseline marker, displacement, force estimate; one block (64 lanes) per frame
__global__ __launch_bounds__(64) void marker_disp_kernel(const int* __restrict__ markers, const int* __restrict__ counts, int max_markers,
                                                         const int* __restrict__ baseline, int n_base, int* __restrict__ disp, double* __restrict__ force) {
  const int f = blockIdx.x, lane = threadIdx.x;
  const int n = min(counts[f], max_markers);
  long long sx = 0, sy = 0;
  for (int e = lane; e < n; e += 64) {
    const int mx = markers[((size_t)f * max_markers + e) * 2], my = markers[((size_t)f * max_markers + e) * 2 + 1];
    long long best = 0x7fffffffffffffffLL; int bi = 0;
    for (int j = 0; j < n_base; ++j) {
      const long long dx = mx - baseline[2 * j], dy = my - baseline[2 * j + 1];
      const long long d2 = dx * dx + dy * dy;
      if (d2 < best) { best = d2; bi = j; }
    }
    const int dx = mx - baseline[2 * bi], dy = my - baseline[2 * bi + 1];
    disp[((size_t)f * max_markers + e) * 2] = dx; disp[((size_t)f * max_markers + e) * 2 + 1] = dy;
    sx += dx; sy += dy;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
  if (lane == 0) {
    double* F = force + (size_t)f * 3;
    if (n == 0 || n_base == 0) { F[0] = F[1] = F[2] = 0.0; return; }
    const double ax = (double)sx / (double)n, ay = (double)sy / (double)n;
    const double mag = sqrt(ax * ax + ay * ay);
    F[0] = mag; F[1] = mag > 0 ? ax / mag : 0.0; F[2] = mag > 0 ? ay / mag : 0.0;
  }
}

}  // namespace

// workspace: binary [N][H][W] u8 | labels [N][H][W] i32 | candidates [N][max_cand] | candidate counters [N]
static size_t al256(size_t b) { return (b + 255) / 256 * 256; }
size_t vt_marker_workspace_bytes(int N, int H, int W, int max_cand) {
  if (N < 1 || H < 1 || W < 1 || max_cand < 1) return 0;
  return al256((size_t)N * H * W) + al256((size_t)N * H * W * 4) + al256((size_t)N * max_cand * sizeof(Cand)) + al256((size_t)N * 4) +
         al256((size_t)N * 256 * 4) + al256((size_t)N * 256);      // + HSR histogram and equalisation table
}

int vt_marker_detect(const uint8_t* frames, int channels, int mode, int N, int H, int W, double min_area, double max_area, int max_cand,
                     int* markers, int* counts, int max_markers, uint8_t* binary_out, void* workspace, vt_stream_t stream) {
  if (!frames || !markers || !counts || !workspace) return vt_fail(VT_ERR_ARG, "vt_marker_detect: null argument");
  if (mode < 0 || (mode & ~0x100) > 2) return vt_fail(VT_ERR_ARG, "vt_marker_detect: mode 0 (standard), 1 (binary input) or 2 (HSR), + 0x100 for the OpenCV <= 3.4.1 gray coefficients");
  const int cfmt = channels | (mode & 0x100);      // gray coefficient set travels with the channel count into the kernels
  mode &= 0xff;
  const int input_is_binary = mode == 1;
  if (input_is_binary && channels != 1) return vt_fail(VT_ERR_ARG, "vt_marker_detect: a binary input has one channel");
  if ((channels != 1 && channels != 3) || N < 1 || H < 5 || W < 5 || max_cand < 1 || max_markers < 1 || (long)H * W >= (1L << 30))
    return vt_fail(VT_ERR_ARG, "vt_marker_detect: bad shape (channels 1|3, H, W >= 5)");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  uint8_t* binary = (uint8_t*)ws; ws += al256((size_t)N * H * W);
  int* labels = (int*)ws; ws += al256((size_t)N * H * W * 4);
  Cand* cand = (Cand*)ws; ws += al256((size_t)N * max_cand * sizeof(Cand));
  int* ncand = (int*)ws; ws += al256((size_t)N * 4);
  int* hist = (int*)ws; ws += al256((size_t)N * 256 * 4);
  uint8_t* lut = (uint8_t*)ws;
  const int HW = H * W;
  // cv::getGaussianKernel(11, sigma <= 0): sigma = 0.3*((11-1)*0.5-1)+0.8 = 2.0, float32 taps
  MeanKernel mk;
  { double k[11], sum = 0; for (int i = 0; i < 11; ++i) { const double x = i - 5.0; k[i] = exp(-(x * x) / (2.0 * 2.0 * 2.0)); sum += k[i]; }
    for (int i = 0; i < 11; ++i) mk.k[i] = (float)(k[i] / sum); }
  if (input_is_binary) hipLaunchKernelGGL(marker_nonzero_kernel, dim3((unsigned)(((long)N * HW + 255) / 256)), dim3(256), 0, s, frames, binary, (long)N * HW);
  else if (mode == 2) {
    if (hipMemsetAsync(hist, 0, (size_t)N * 256 * 4, s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "vt_marker_detect: memset");
    hipLaunchKernelGGL(marker_hist_kernel, dim3(32, N), dim3(256), 0, s, frames, cfmt, H, W, hist);
    hipLaunchKernelGGL(marker_lut_kernel, dim3(N), dim3(64), 0, s, (const int*)hist, HW, lut);
    hipLaunchKernelGGL(marker_binary_kernel<true>, dim3((W + TS - 1) / TS, (H + TS - 1) / TS, N), dim3(256), 0, s, frames, cfmt, H, W, mk, (const uint8_t*)lut, binary);
  } else hipLaunchKernelGGL(marker_binary_kernel<false>, dim3((W + TS - 1) / TS, (H + TS - 1) / TS, N), dim3(256), 0, s, frames, cfmt, H, W, mk, (const uint8_t*)nullptr, binary);
  hipLaunchKernelGGL(marker_label_local_init_kernel, dim3((HW + 255) / 256, N), dim3(256), 0, s, (const uint8_t*)binary, labels, HW);
  hipLaunchKernelGGL(marker_label_merge_kernel, dim3((HW + 255) / 256, N), dim3(256), 0, s, (const uint8_t*)binary, labels, H, W);
  if (hipMemsetAsync(ncand, 0, (size_t)N * 4, s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "vt_marker_detect: memset");
  // after the merge pass every pixel's chain ends at its component's smallest index; roots are exactly the pixels with L[i] == i
  hipLaunchKernelGGL(marker_trace_kernel, dim3((HW + 255) / 256, N), dim3(256), 0, s, (const uint8_t*)binary, (const int*)labels, H, W, min_area, max_area, cand,
                     ncand, max_cand);
  hipLaunchKernelGGL(marker_order_kernel, dim3(N), dim3(256), 0, s, (const Cand*)cand, ncand, max_cand, markers, counts, max_markers);
  if (binary_out && hipMemcpyAsync(binary_out, binary, (size_t)N * HW, hipMemcpyDeviceToDevice, s) != hipSuccess) return vt_fail(VT_ERR_LAUNCH, "vt_marker_detect: copy");
  return vt_check_launch();
}

int vt_marker_displacement(const int* markers, const int* counts, int N, int max_markers, const int* baseline, int n_base, int* disp, double* force,
                           vt_stream_t stream) {
  if (!markers || !counts || !disp || !force || (n_base > 0 && !baseline) || N < 1 || max_markers < 1 || n_base < 0)
    return vt_fail(VT_ERR_ARG, "vt_marker_displacement: bad argument");
  hipLaunchKernelGGL(marker_disp_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, markers, counts, max_markers, baseline, n_base, disp, force);
  return vt_check_launch();
}
