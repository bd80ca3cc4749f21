"""Oracle: GelSight marker tracker -> marker displacements / force estimate m_t (test infrastructure; see oracle/__init__.py).

Restates /root/reference/VLA/residual_controller/tactile/marker/marker_tracker.py:
  init_standard :81-114 (BGR2GRAY -> GaussianBlur 5x5 -> adaptiveThreshold(GAUSSIAN_C, BINARY_INV, 11, 2) -> MORPH_OPEN 3x3),
  init_HSR :116-152 (BGR2GRAY -> invert -> equalizeHist -> GaussianBlur 5x5 -> threshold 50 -> MORPH_OPEN 3x3),
  detect_markers :154-241 (findContours EXTERNAL -> contourArea in (10, 500) -> contour moments -> int centroid),
  match_and_compute_displacement :308-341 (nearest baseline marker), estimate_force :343-373 (mean displacement, norm, direction).

The image primitives are OpenCV's (`cv2`, third party, absent from /root/reference and from this image, not version-pinned
by the reference) — PARITY UNPINNED for those: they are restated here from OpenCV's published algorithms (4.x sources):
  * cvtColor BGR2GRAY 8-bit, TWO coefficient sets (GRAY_COEFFS, `gray_mode`): "cv4" = OpenCV >= 3.4.2 and every 4.x
    (modules/imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar>: BY15 = 3735, GY15 = 19235, RY15 = 9798, gray_shift = 15,
    CV_DESCALE -> (B*3735 + G*19235 + R*9798 + 16384) >> 15) — the DEFAULT, since the reference's unpinned `opencv-python`
    (octopi/requirements.txt:2) resolves to 4.x; "cv3" = OpenCV <= 3.4.1 (color.cpp, the 14-bit table B2Y = 1868, G2Y = 9617,
    R2Y = 4899, yuv_shift = 14 -> (B*1868 + G*9617 + R*4899 + 8192) >> 14).  The two differ by one grey level on ~0.3 % of
    random colour pixels (and never on grey ones, B = G = R); known-answer tests for both in tests/test_marker.py.
  * GaussianBlur 5x5, sigma 0, 8-bit: fixed kernel [1 4 6 4 1]/16 (small_gaussian_tab), fixed-point, round half up,
    BORDER_REFLECT_101
  * adaptiveThreshold GAUSSIAN_C: float GaussianBlur 11x11 (sigma = 0.3*((11-1)*0.5-1)+0.8 = 2.0), BORDER_REPLICATE, rounded
    half-to-even to 8 bit; THRESH_BINARY_INV: dst = 255 where src - mean <= -floor(C)
  * morphologyEx OPEN with a 3x3 rectangle = erode then dilate (border = +inf / -inf: never wins)
  * findContours RETR_EXTERNAL (Suzuki-Abe border following of 8-connected components, outer borders only), contours
    returned last-found-first; contourArea / moments = Green's-theorem polygon integrals over the border pixel centres
    (CHAIN_APPROX_SIMPLE only drops collinear points, which contribute nothing).
The composition (calibration flow, filtering, matching, force) is pinned to the reference class itself, driven through these
primitives by tools/make_golden_marker.py -> tests/golden/g12_marker.npz.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


# ------------------------------------------------------------------ OpenCV primitives (restated)
GRAY_COEFFS = {"cv4": (3735, 19235, 9798, 15), "cv3": (1868, 9617, 4899, 14)}     # (B, G, R, shift)
GRAY_MODE = "cv4"                                                                  # module default (see the header)


def bgr2gray(frame: np.ndarray, gray_mode: str = None) -> np.ndarray:
    cb, cg, cr, sh = GRAY_COEFFS[gray_mode or GRAY_MODE]
    f = frame.astype(np.int64)
    return ((f[..., 0] * cb + f[..., 1] * cg + f[..., 2] * cr + (1 << (sh - 1))) >> sh).astype(np.uint8)


def _reflect101(a: np.ndarray, r: int) -> np.ndarray:
    return np.pad(a, r, mode="reflect")


def gaussian_blur5_u8(gray: np.ndarray) -> np.ndarray:
    k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    p = _reflect101(gray.astype(np.int64), 2)
    H, W = gray.shape
    h = sum(k[i] * p[:, i:i + W] for i in range(5))                 # horizontal, x16
    v = sum(k[i] * h[i:i + H, :] for i in range(5))                 # vertical, x256
    return ((v + 128) >> 8).astype(np.uint8)


def gaussian_kernel(ksize: int, sigma: float = 0.0) -> np.ndarray:
    """cv::getGaussianKernel for float images (no fixed table: sigma from ksize when <= 0), float32 like OpenCV."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float32)


def adaptive_threshold_gaussian_inv(gray: np.ndarray, block: int = 11, C: float = 2.0) -> np.ndarray:
    k = gaussian_kernel(block)
    r = block // 2
    p = np.pad(gray.astype(np.float32), r, mode="edge")              # BORDER_REPLICATE
    H, W = gray.shape
    h = np.zeros((H + 2 * r, W), dtype=np.float32)
    for i in range(block):
        h += k[i] * p[:, i:i + W]
    v = np.zeros((H, W), dtype=np.float32)
    for i in range(block):
        v += k[i] * h[i:i + H, :]
    mean = np.clip(np.rint(v), 0, 255).astype(np.int32)              # saturate_cast<uchar>(float): round half to even
    idelta = int(np.floor(C))                                        # THRESH_BINARY_INV uses cvFloor(delta)
    return np.where(gray.astype(np.int32) - mean <= -idelta, 255, 0).astype(np.uint8)


def morph_open3(binary: np.ndarray) -> np.ndarray:
    H, W = binary.shape
    p = np.pad(binary, 1, mode="constant", constant_values=255)
    er = np.minimum.reduce([p[i:i + H, j:j + W] for i in range(3) for j in range(3)])
    p = np.pad(er, 1, mode="constant", constant_values=0)
    return np.maximum.reduce([p[i:i + H, j:j + W] for i in range(3) for j in range(3)])


# 8-neighbourhood in clockwise order starting East (x right, y down): E, SE, S, SW, W, NW, N, NE
_NB = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]


def external_contours(binary: np.ndarray) -> List[np.ndarray]:
    """Outer borders of the 8-connected components of the non-zero pixels (Suzuki-Abe, RETR_EXTERNAL), each as the closed
    sequence of border pixel centres [(x, y), ...]; a component's trace starts at its first pixel in raster order (which has
    background to its west and north) and runs with the component on its right-hand side... the orientation does not matter
    to |area| and the centroid.  Returned last-found-first like cv2.findContours."""
    img = (binary != 0)
    H, W = img.shape
    lab = np.zeros((H, W), dtype=np.int32)
    out: List[np.ndarray] = []
    # component labelling by flood fill (which pixels belong to an already traced component)
    for y0 in range(H):
        row = img[y0]
        for x0 in range(W):
            if not row[x0] or lab[y0, x0]:
                continue
            cid = len(out) + 1
            stack = [(x0, y0)]
            lab[y0, x0] = cid
            while stack:
                x, y = stack.pop()
                for dx, dy in _NB:
                    xx, yy = x + dx, y + dy
                    if 0 <= xx < W and 0 <= yy < H and img[yy, xx] and not lab[yy, xx]:
                        lab[yy, xx] = cid
                        stack.append((xx, yy))
            out.append(_trace_outer(img, x0, y0))
    return out[::-1]


def _trace_outer(img: np.ndarray, x0: int, y0: int) -> np.ndarray:
    """Moore-neighbour border following from the raster-first pixel (x0, y0) of a component (its W, NW, N, NE neighbours are
    background), visiting the outer border pixels in order; stops with Jacob's criterion (back at the start, entering as at first)."""
    H, W = img.shape

    def fg(x, y):
        return 0 <= x < W and 0 <= y < H and img[y, x]

    pts = [(x0, y0)]
    # first move: search clockwise starting from the West neighbour (index 4) -> NW, N, NE, E, ...
    d = 4
    cur = (x0, y0)
    first_dir = None
    while True:
        found = False
        for i in range(8):
            k = (d + i) % 8
            nx, ny = cur[0] + _NB[k][0], cur[1] + _NB[k][1]
            if fg(nx, ny):
                found = True
                break
        if not found:                       # isolated pixel
            break
        if cur == (x0, y0):
            if first_dir is None:
                first_dir = k
            elif k == first_dir and len(pts) > 1:
                break
        cur = (nx, ny)
        pts.append(cur)
        d = (k + 5) % 8                     # restart the clockwise search just past the pixel we came from
    if len(pts) > 1 and pts[-1] == pts[0]:
        pts.pop()
    return np.array(pts, dtype=np.int64)


def contour_moments(pts: np.ndarray) -> Tuple[float, float, float]:
    """(m00, m10, m01) of the polygon through `pts` (cv::contourMoments, Green's theorem); m00 sign-normalised like OpenCV."""
    n = len(pts)
    if n == 0:
        return 0.0, 0.0, 0.0
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    xp, yp = np.roll(x, 1), np.roll(y, 1)
    dxy = xp * y - x * yp
    a00 = dxy.sum()
    a10 = (dxy * (xp + x)).sum()
    a01 = (dxy * (yp + y)).sum()
    if abs(a00) < 1e-12:
        return 0.0, 0.0, 0.0
    if a00 > 0:
        return a00 * 0.5, a10 / 6.0, a01 / 6.0
    return -a00 * 0.5, -a10 / 6.0, -a01 / 6.0


def contour_area(pts: np.ndarray) -> float:
    return contour_moments(pts)[0]


# ------------------------------------------------------------------ the reference's pipeline (marker_tracker.py)
def preprocess_standard(frame: np.ndarray, gray_mode: str = None) -> np.ndarray:
    """init_standard :81-114."""
    gray = bgr2gray(frame, gray_mode) if frame.ndim == 3 else frame
    return morph_open3(adaptive_threshold_gaussian_inv(gaussian_blur5_u8(gray), 11, 2))


def equalize_hist(gray: np.ndarray) -> np.ndarray:
    """cv::equalizeHist: lut[i] = saturate_cast<uchar>(cumsum over (first non-empty bin, i] * 255 / (total - hist[first])), float32
    scale, round half to even; a constant image is returned unchanged."""
    hist = np.bincount(gray.reshape(-1), minlength=256).astype(np.int64)
    total = int(gray.size)
    i0 = int(np.nonzero(hist)[0][0])
    if hist[i0] == total:
        return gray.copy()
    scale = np.float32(255.0) / np.float32(total - hist[i0])
    lut = np.zeros(256, dtype=np.uint8)
    s = 0
    for i in range(i0 + 1, 256):
        s += int(hist[i])
        lut[i] = np.uint8(min(255, max(0, int(np.rint(np.float32(s) * scale)))))
    return lut[gray]


def preprocess_hsr(frame: np.ndarray, gray_mode: str = None) -> np.ndarray:
    """init_HSR :116-152: gray -> 255 - gray -> equalizeHist -> GaussianBlur 5x5 -> threshold(> 50) -> MORPH_OPEN 3x3."""
    gray = bgr2gray(frame, gray_mode) if frame.ndim == 3 else frame
    eq = equalize_hist((255 - gray.astype(np.int32)).astype(np.uint8))
    return morph_open3(np.where(gaussian_blur5_u8(eq) > 50, 255, 0).astype(np.uint8))


def detect_markers(processed: np.ndarray, min_area: float = 10, max_area: float = 500) -> np.ndarray:
    """detect_markers :154-183 (the exact-count / too-few branches; the KMeans branch for surplus detections is host logic)."""
    markers = []
    for c in external_contours(processed):
        m00, m10, m01 = contour_moments(c)
        if min_area < m00 < max_area and m00 != 0:
            markers.append([int(m10 / m00), int(m01 / m00)])
    return np.array(markers, dtype=np.int64).reshape(-1, 2)


def match_displacement(current: np.ndarray, baseline: np.ndarray) -> np.ndarray:
    """match_and_compute_displacement :308-341: each current marker minus its nearest baseline marker (cKDTree.query k=1;
    ties resolve to the lower baseline index here)."""
    if len(current) == 0:
        return np.zeros((0, 2), dtype=np.int64)
    d2 = ((current[:, None, :] - baseline[None, :, :]) ** 2).sum(-1)
    return current - baseline[d2.argmin(1)]


def estimate_force(displacement: np.ndarray) -> Tuple[float, np.ndarray]:
    """estimate_force :343-373."""
    if len(displacement) == 0:
        return 0.0, np.zeros(2)
    avg = displacement.astype(np.float64).mean(0)
    mag = float(np.linalg.norm(avg))
    return mag, (avg / mag if mag > 0 else np.zeros(2))


def synth_gel_frame(*args, **kw) -> np.ndarray:
    """The synthetic GelSight frame generator lives with the other synthetic inputs (vlatouch/synth.py); kept here as an alias for
    the golden-vector script and the tests."""
    from vlatouch import synth
    return synth.synth_gel_frame(*args, **kw)
