"""GelSight marker tracker.  Golden g12 = outputs of the REFERENCE class
(/root/reference/VLA/residual_controller/tactile/marker/marker_tracker.py::EnhancedMarkerTracker) driven in the build container
through a stand-in `cv2` module made of oracle/marker.py's restated OpenCV primitives (tools/make_golden_marker.py): it pins
the class logic; the primitives themselves are third party and unpinned (oracle/marker.py header).
  CPU: the oracle's own composition == the reference class's outputs; primitive sanity checks.
  GPU: the device tracker == golden / oracle, bit-exact (integer and fp64 work), incl. the batched stream API and edge cases."""
import numpy as np
import pytest
import torch

from tests import cases
from oracle import marker as M
from tools.make_golden_marker import frames as golden_frames, hsr_frames, MOTION


def G():
    return np.load(f"{cases.GOLDEN}/g12_marker.npz")


def test_oracle_matches_reference_class_outputs():
    g, fr = G(), golden_frames()
    base = M.detect_markers(M.preprocess_standard(fr[0]))
    assert np.array_equal(base, g["baseline"]) and len(base) == 63
    for i, f in enumerate(fr):
        proc = M.preprocess_standard(f)
        assert int(proc.astype(np.int64).sum()) == int(g[f"binary_sum_{i}"])
        cur = M.detect_markers(proc)
        assert np.array_equal(cur, g[f"markers_{i}"])
        disp = M.match_displacement(cur, base)
        assert np.array_equal(disp, g[f"disp_{i}"])
        mag, d = M.estimate_force(disp)
        assert np.allclose([mag, d[0], d[1]], g[f"force_{i}"], rtol=0, atol=1e-12)
    # the synthetic motion is recovered to about a pixel (sanity of the whole chain, not a parity statement)
    for i, ((sx, sy), bulge) in enumerate(MOTION):
        mean = g[f"disp_{i}"].mean(0)
        assert abs(mean[0] - sx) < 1.0 and abs(mean[1] - sy) < 1.0, (i, mean)


def test_oracle_hsr_matches_reference_class_outputs():
    g, fr = G(), hsr_frames()
    base = M.detect_markers(M.preprocess_hsr(fr[0]))
    assert np.array_equal(base, g["hsr_baseline"]) and len(base) == 63
    for i, f in enumerate(fr):
        proc = M.preprocess_hsr(f)
        assert int(proc.astype(np.int64).sum()) == int(g[f"hsr_binary_sum_{i}"])
        cur = M.detect_markers(proc)
        assert np.array_equal(cur, g[f"hsr_markers_{i}"])
        assert np.array_equal(M.match_displacement(cur, base), g[f"hsr_disp_{i}"])
    # equalizeHist known answers: a constant image is unchanged; two equal-population levels map to 0 and 255
    assert (M.equalize_hist(np.full((4, 4), 9, np.uint8)) == 9).all()
    two = np.array([[10, 10, 200, 200]], np.uint8)
    assert M.equalize_hist(two).tolist() == [[0, 0, 255, 255]]


def test_oracle_primitives_known_answers():
    # BGR -> gray fixed point: pure colours (OpenCV documents Y = 0.299 R + 0.587 G + 0.114 B)
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], dtype=np.uint8)
    assert M.bgr2gray(px).tolist() == [[29, 150, 76, 255]]
    assert M.bgr2gray(px, "cv3").tolist() == [[29, 150, 76, 255]]
    # the two published coefficient sets (15-bit, OpenCV >= 3.4.2 / 4.x — the default — and 14-bit, OpenCV <= 3.4.1), on pixels whose exact
    # luma 0.299 R + 0.587 G + 0.114 B sits within 2e-3 of a half: 27.499, 128.502, 133.501, 103.499
    edge = np.array([[[55, 0, 71], [63, 197, 19], [236, 122, 117], [111, 59, 188]]], dtype=np.uint8)
    assert M.bgr2gray(edge, "cv4").tolist() == [[27, 129, 133, 103]] == M.bgr2gray(edge).tolist()
    assert M.bgr2gray(edge, "cv3").tolist() == [[28, 128, 134, 104]]
    g3 = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)            # grey pixels: both sets are exact
    assert (M.bgr2gray(g3, "cv4") == g3[..., 0]).all() and (M.bgr2gray(g3, "cv3") == g3[..., 0]).all()
    # binomial blur of an impulse of 256 -> the [1 4 6 4 1]^2 / 256 kernel itself (x256/256 rounded), constant stays constant
    imp = np.zeros((9, 9), dtype=np.uint8); imp[4, 4] = 255
    b = M.gaussian_blur5_u8(imp)
    assert b[4, 4] == (36 * 255 + 128) >> 8 and b[4, 2] == (6 * 255 + 128) >> 8 and b[2, 2] == (255 + 128) >> 8
    assert (M.gaussian_blur5_u8(np.full((7, 9), 93, np.uint8)) == 93).all()
    # polygon moments: 5 x 3 pixel rectangle -> contour through pixel centres has area 4 x 2 and centroid at its middle
    r = np.zeros((8, 10), np.uint8); r[2:5, 3:8] = 255
    (c,) = M.external_contours(r)
    m00, m10, m01 = M.contour_moments(c)
    assert m00 == 8.0 and m10 / m00 == 5.0 and m01 / m00 == 3.0
    # opening removes a lone pixel and a 2-wide bar, keeps a 3x3 block
    o = np.zeros((12, 12), np.uint8); o[1, 1] = 255; o[4:6, 2:9] = 255; o[8:11, 8:11] = 255
    assert M.morph_open3(o).sum() == 9 * 255
    # diagonal pixels are ONE 8-connected component
    dg = np.zeros((6, 6), np.uint8); dg[1, 1] = dg[2, 2] = dg[3, 3] = 255
    assert len(M.external_contours(dg)) == 1


# ------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


def tracker():
    from residual_controller.tactile.marker.marker_tracker import EnhancedMarkerTracker
    return EnhancedMarkerTracker(grid_rows=7, grid_cols=9, device="cuda:0")


@gpu
def test_device_tracker_matches_reference_golden():
    g, fr = G(), golden_frames()
    tr = tracker()
    base = tr.calibrate(fr[0])
    assert np.array_equal(base, g["baseline"])
    assert np.allclose(tr.ideal_grid, g["ideal_grid"])
    for i, f in enumerate(fr):
        proc = tr.preprocess_frame(f)
        assert proc.dtype == np.uint8 and int(proc.astype(np.int64).sum()) == int(g[f"binary_sum_{i}"])
        assert np.array_equal(proc, M.preprocess_standard(f))
        cur = tr.detect_markers(proc)
        assert np.array_equal(cur, g[f"markers_{i}"])
        disp = tr.get_marker_state(f)
        assert np.array_equal(disp, g[f"disp_{i}"])
        mag, d = tr.estimate_force(disp)
        assert np.allclose([mag, d[0], d[1]], g[f"force_{i}"], rtol=0, atol=1e-12)


@gpu
def test_device_tracker_hsr_matches_reference_golden():
    from residual_controller.tactile.marker.marker_tracker import EnhancedMarkerTracker
    g, fr = G(), hsr_frames()
    tr = EnhancedMarkerTracker(grid_rows=7, grid_cols=9, gelsight_version='HSR', device="cuda:0")
    assert np.array_equal(tr.calibrate(fr[0]), g["hsr_baseline"])
    for i, f in enumerate(fr):
        proc = tr.preprocess_frame(f)
        assert int(proc.astype(np.int64).sum()) == int(g[f"hsr_binary_sum_{i}"]) and np.array_equal(proc, M.preprocess_hsr(f))
        assert np.array_equal(tr.detect_markers(proc), g[f"hsr_markers_{i}"])
        assert np.array_equal(tr.get_marker_state(f), g[f"hsr_disp_{i}"])
    tr2 = EnhancedMarkerTracker(grid_rows=7, grid_cols=9, gelsight_version='HSR', device="cuda:0")
    disp, mag, _ = tr2.track_frames(np.stack(fr))
    for i in range(len(fr)):
        assert np.array_equal(disp[i, :int(tr2.last_counts[i])], g[f"hsr_disp_{i}"])


@gpu
def test_device_tracker_batched_stream_equals_frame_by_frame():
    g, fr = G(), golden_frames()
    tr = tracker()
    disp, mag, direction = tr.track_frames(np.stack(fr))
    assert np.array_equal(tr.baseline_markers, g["baseline"])
    for i in range(len(fr)):
        n = int(tr.last_counts[i])
        assert np.array_equal(disp[i, :n], g[f"disp_{i}"])
        assert np.allclose([mag[i], direction[i, 0], direction[i, 1]], g[f"force_{i}"], rtol=0, atol=1e-12)


@gpu
@pytest.mark.parametrize("H,W", [(240, 320), (97, 131), (33, 40)])
def test_device_tracker_vs_oracle_random_blobs_and_edges(H, W):
    """Ragged sizes (not multiples of the 32-pixel tile), blobs touching the image border, specks under / over the area
    filter, a blank frame, a gray (1-channel) frame; against the oracle run live."""
    rng = np.random.default_rng(H * 1000 + W)
    tr = tracker()
    frames = []
    for k in range(4):
        f = np.full((H, W, 3), 170, np.float64) + rng.normal(0, 3, (H, W, 3))
        yy, xx = np.mgrid[0:H, 0:W]
        n = 0 if k == 3 else 12
        for _ in range(n):
            cx, cy, r = rng.uniform(-2, W + 2), rng.uniform(-2, H + 2), rng.uniform(1.0, 9.0)
            f *= (1 - 0.8 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))[..., None]
        frames.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
    for f in frames:
        proc = tr.preprocess_frame(f)
        ref = M.preprocess_standard(f)
        assert np.array_equal(proc, ref)
        tr.expected_markers = 10 ** 6                     # the raw candidate list, whatever its length
        got = tr.detect_markers(proc)
        want = M.detect_markers(ref)
        assert np.array_equal(np.asarray(got).reshape(-1, 2), want)
        # the contour stage alone, from a foreign binary image
        again = tr.detect_markers(ref.copy())
        assert np.array_equal(np.asarray(again).reshape(-1, 2), want)
    gray = M.bgr2gray(frames[0])
    assert np.array_equal(tr.preprocess_frame(gray), M.preprocess_standard(gray))
    tr2 = tracker()
    with pytest.raises(ValueError):
        tr2.preprocess_frame(frames[0].astype(np.float32))
    # the HSR variant on arbitrary frames (incl. its degenerate "whole gel is foreground" outcome) == the oracle
    tr2.gelsight_version = "HSR"
    tr2.expected_markers = 10 ** 6
    for f in (frames[0], frames[3], np.full((H, W, 3), 77, np.uint8)):
        ref = M.preprocess_hsr(f)
        assert np.array_equal(tr2.preprocess_frame(f), ref)
        assert np.array_equal(np.asarray(tr2.detect_markers(tr2.preprocess_frame(f))).reshape(-1, 2), M.detect_markers(ref))


@gpu
@pytest.mark.parametrize("gray", ["cv4", "cv3"])
def test_device_gray_coefficient_sets(gray):
    """BGR2GRAY on the device with either OpenCV coefficient set == the oracle's, on colour noise (where the sets disagree on ~0.3 % of the
    pixels) and on the four known-answer pixels; the binary image that follows (blur, adaptive threshold, open) likewise."""
    from residual_controller.tactile.marker.marker_tracker import EnhancedMarkerTracker
    rng = np.random.default_rng(5)
    f = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    f[0, :4] = [[55, 0, 71], [63, 197, 19], [236, 122, 117], [111, 59, 188]]
    assert (M.bgr2gray(f, "cv4") != M.bgr2gray(f, "cv3")).any()
    tr = EnhancedMarkerTracker(7, 9, device="cuda", opencv_gray=gray)
    assert np.array_equal(tr.preprocess_frame(f), M.preprocess_standard(f, gray))
    tr.gelsight_version = "HSR"
    assert np.array_equal(tr.preprocess_frame(f), M.preprocess_hsr(f, gray))
    with pytest.raises(ValueError):
        EnhancedMarkerTracker(7, 9, device="cuda", opencv_gray="cv2")


@gpu
def test_device_displacement_ties_empty_and_force():
    tr = tracker()
    tr.baseline_markers = np.array([[10, 10], [20, 10], [10, 20]], dtype=np.int64)
    cur = np.array([[15, 10], [11, 21], [40, 40]], dtype=np.int64)          # first marker is equidistant to baseline 0 and 1 -> lower index
    d = tr.match_and_compute_displacement(cur)
    assert np.array_equal(d, M.match_displacement(cur, tr.baseline_markers))
    assert len(tr.match_and_compute_displacement(np.zeros((0, 2), np.int64))) == 0
    mag, d = tr.estimate_force(np.array([]))
    assert mag == 0 and np.array_equal(d, [0, 0])
