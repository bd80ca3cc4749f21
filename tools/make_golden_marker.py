#!/usr/bin/env python
"""Golden vectors for the GelSight marker tracker: runs the REFERENCE class
/root/reference/VLA/residual_controller/tactile/marker/marker_tracker.py::EnhancedMarkerTracker in this container.
`cv2` is absent (and un-pinned by the reference): a stand-in MODULE exposes the handful of OpenCV primitives the class
calls, implemented by oracle/marker.py's restatements of OpenCV's published algorithms (so the primitives are "parity
unpinned", the class logic — calibration flow, area filter, centroid truncation, nearest-baseline matching, force
estimate — is the reference's own code).  Inputs are regenerated from seeds (oracle.marker.synth_gel_frame); only the
reference's OUTPUTS are stored:  tests/golden/g12_marker.npz.
    python tools/make_golden_marker.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vla-touch_amd")]
from oracle import marker as M  # noqa: E402

REF = "/root/reference/VLA/residual_controller/tactile/marker/marker_tracker.py"
SEED, NFRAMES = 2024, 6
MOTION = [((0.0, 0.0), 0.0), ((1.6, -0.8), 0.0), ((3.2, 1.4), 2.5), ((-2.4, 2.2), 5.0), ((0.7, -3.1), 7.5), ((4.0, 3.0), 3.0)]


def frames():
    rng = np.random.default_rng(SEED)
    return [M.synth_gel_frame(rng, shift=s, bulge=b) for s, b in MOTION]


def hsr_frames():
    """Frames for the HSR variant (init_HSR inverts, equalises the histogram and thresholds at 50): a UNIFORM bright gel — the one
    background histogram equalisation maps to 0 — with dark markers of varied depth, noise-free; rigid shifts between frames."""
    out = []
    H, W = 240, 320
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for (sx, sy), _ in MOTION[:4]:
        img = np.full((H, W), 200.0)
        k = 0
        for y in np.linspace(24, H - 24, 7):
            for x in np.linspace(28, W - 28, 9):
                d2 = (xx - x - sx) ** 2 + (yy - y - sy) ** 2
                img *= 1 - (0.55 + 0.04 * (k % 7)) * np.exp(-d2 / (2 * (2.4 + 0.2 * (k % 5)) ** 2))
                k += 1
        g = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        out.append(np.stack([g, g, g], axis=-1))
    return out


def cv2_standin():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2GRAY, cv2.ADAPTIVE_THRESH_GAUSSIAN_C, cv2.THRESH_BINARY_INV, cv2.THRESH_BINARY = 6, 1, 1, 0
    cv2.MORPH_OPEN, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE = 2, 0, 2

    def cvtColor(img, code):
        assert code == cv2.COLOR_BGR2GRAY
        return M.bgr2gray(img)

    def GaussianBlur(img, ksize, sigma):
        assert tuple(ksize) == (5, 5) and sigma == 0 and img.dtype == np.uint8
        return M.gaussian_blur5_u8(img)

    def adaptiveThreshold(img, maxval, method, ttype, block, C):
        assert maxval == 255 and method == cv2.ADAPTIVE_THRESH_GAUSSIAN_C and ttype == cv2.THRESH_BINARY_INV
        return M.adaptive_threshold_gaussian_inv(img, block, C)

    def equalizeHist(img):
        return M.equalize_hist(img)

    def threshold(img, thresh, maxval, ttype):
        assert ttype == cv2.THRESH_BINARY and maxval == 255
        return thresh, np.where(img > thresh, 255, 0).astype(np.uint8)

    def morphologyEx(img, op, kernel):
        assert op == cv2.MORPH_OPEN and kernel.shape == (3, 3) and kernel.all()
        return M.morph_open3(img)

    def findContours(img, mode, method):
        assert mode == cv2.RETR_EXTERNAL
        return [c.reshape(-1, 1, 2).astype(np.int32) for c in M.external_contours(img)], None

    def contourArea(c):
        return M.contour_area(c.reshape(-1, 2))

    def moments(c):
        m00, m10, m01 = M.contour_moments(c.reshape(-1, 2))
        return {"m00": m00, "m10": m10, "m01": m01}

    for f in (cvtColor, GaussianBlur, adaptiveThreshold, morphologyEx, findContours, contourArea, moments, equalizeHist, threshold):
        setattr(cv2, f.__name__, f)
    return cv2


def main():
    sys.modules["cv2"] = cv2_standin()
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, mpl.pyplot
    spec = importlib.util.spec_from_file_location("ref_marker_tracker", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fr = frames()
    tr = mod.EnhancedMarkerTracker(grid_rows=7, grid_cols=9)
    base = tr.calibrate(fr[0])
    out = {"baseline": np.asarray(base), "ideal_grid": np.asarray(tr.ideal_grid)}
    for i, f in enumerate(fr):
        proc = tr.preprocess_frame(f)
        cur = tr.detect_markers(proc)
        disp = tr.get_marker_state(f)
        mag, direction = tr.estimate_force(disp)
        out[f"binary_sum_{i}"] = np.int64(proc.astype(np.int64).sum())
        out[f"markers_{i}"] = np.asarray(cur)
        out[f"disp_{i}"] = np.asarray(disp)
        out[f"force_{i}"] = np.array([mag, direction[0], direction[1]], dtype=np.float64)
        print(i, len(cur), mag, direction)
    # the 'HSR' sensor variant (init_HSR): dark gel, bright markers (its inversion + equalisation + fixed threshold expect that)
    hf = hsr_frames()
    th = mod.EnhancedMarkerTracker(grid_rows=7, grid_cols=9, gelsight_version='HSR')
    out["hsr_baseline"] = np.asarray(th.calibrate(hf[0]))
    for i, f in enumerate(hf):
        proc = th.preprocess_frame(f)
        out[f"hsr_binary_sum_{i}"] = np.int64(proc.astype(np.int64).sum())
        out[f"hsr_markers_{i}"] = np.asarray(th.detect_markers(proc))
        disp = th.get_marker_state(f)
        out[f"hsr_disp_{i}"] = np.asarray(disp)
        print("HSR", i, len(out[f"hsr_markers_{i}"]), th.estimate_force(disp)[0])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g12_marker.npz"), **out)


if __name__ == "__main__":
    main()
