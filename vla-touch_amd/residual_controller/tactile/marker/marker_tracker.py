"""GelSight marker tracker — mirror of the reference's
VLA/residual_controller/tactile/marker/marker_tracker.py:8-373 (`EnhancedMarkerTracker`, 'standard' sensor).

Same constructor, attributes (`grid_dims`, `expected_markers`, `baseline_markers`, `ideal_grid`) and methods: `calibrate`,
`preprocess_frame`, `detect_markers`, `create_ideal_grid`, `get_marker_state`, `match_and_compute_displacement`,
`estimate_force`, returning numpy arrays like the reference.  The image chain (gray, blur, adaptive threshold, open), the
connected components / outer-contour moments and the nearest-baseline matching run on the MI355X (csrc/vt_marker.hip through
vt_marker_detect / vt_marker_displacement); there is no cv2 and no CPU path.

MI355X-first addition: `track_frames(frames[N, H, W, 3])` labels a whole GelSight stream (an episode) in one batched call —
the reference walks it frame by frame (`process_image_sequence`, :376-450).

Both sensor variants run on the device: 'standard' (`init_standard`, :81-114) and 'HSR' (`init_HSR`, :116-152: invert, histogram
equalisation, blur, fixed threshold).  Kept on the host, as in the reference: the KMeans refinement when MORE than
`expected_markers` blobs survive the area filter (:205-229, sklearn, a rare branch on a handful of points) and `filter_coords`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from vlatouch import _lib as L
from vlatouch.engine import _Workspace

MAX_CAND = 1024          # components per frame that may pass the area filter before the host sees an overflow


class EnhancedMarkerTracker:
    def __init__(self, grid_rows=7, grid_cols=9, calibration_frame=None, gelsight_version='standard', device="cuda", opencv_gray="cv4"):
        """opencv_gray (not in the reference, whose cv2 is unpinned): which cv2.cvtColor(BGR2GRAY) arithmetic the device restates — "cv4" =
        OpenCV >= 3.4.2 / 4.x (15-bit coefficients, what `pip install opencv-python` gives today), "cv3" = OpenCV <= 3.4.1 (14-bit)."""
        if opencv_gray not in ("cv4", "cv3"):
            raise ValueError(f"opencv_gray must be 'cv4' or 'cv3', got {opencv_gray!r}")
        self.opencv_gray = opencv_gray
        self.grid_dims = (grid_rows, grid_cols) if grid_rows and grid_cols else None
        self.expected_markers = grid_rows * grid_cols
        self.baseline_markers = None
        self.gelsight_version = gelsight_version
        self.device = L.require_gpu(device)
        self._ws = _Workspace(self.device)
        if calibration_frame is not None:
            print("Calibrate with a reference frame.")
            self.calibrate(calibration_frame)

    # ------------------------------------------------------------------ device plumbing
    def _frames(self, frames) -> torch.Tensor:
        """-> [N, H, W, C] uint8 on the device (C = 3 BGR or 1 gray); accepts one frame [H, W(, 3)] or a batch."""
        t = torch.from_numpy(np.ascontiguousarray(frames)) if isinstance(frames, np.ndarray) else frames
        if t.dtype != torch.uint8:
            raise ValueError(f"frames must be uint8 (cv2 images), got {t.dtype}")
        if t.dim() == 2:
            t = t[None, :, :, None]
        elif t.dim() == 3:
            t = t[None] if t.shape[-1] == 3 else t[..., None]
        if t.dim() != 4 or t.shape[-1] not in (1, 3):
            raise ValueError(f"expected [H, W], [H, W, 3], [N, H, W] or [N, H, W, 3] frames, got {tuple(t.shape)}")
        return t.to(self.device).contiguous()

    def _detect(self, frames: torch.Tensor, min_area=10.0, max_area=500.0, want_binary=False, is_binary=False):
        """frames [N,H,W,C] uint8 on device -> (markers [N, max, 2] int32, counts [N] int32, binary [N,H,W] uint8 | None)."""
        N, H, W, Cc = frames.shape
        max_markers = max(4 * self.expected_markers, 256)
        markers = torch.zeros(N, max_markers, 2, dtype=torch.int32, device=self.device)
        counts = torch.zeros(N, dtype=torch.int32, device=self.device)
        binary = torch.empty(N, H, W, dtype=torch.uint8, device=self.device) if want_binary else None
        lib = L.lib()
        ws = self._ws.get(lib.vt_marker_workspace_bytes(N, H, W, MAX_CAND))
        mode = (1 if is_binary else (2 if self.gelsight_version == 'HSR' else 0)) | (0x100 if self.opencv_gray == "cv3" else 0)
        L.check(lib.vt_marker_detect(L.ptr(frames), Cc, mode, N, H, W, float(min_area), float(max_area), MAX_CAND, L.ptr(markers), L.ptr(counts),
                                     max_markers, L.ptr(binary), L.ptr(ws), L.stream_ptr(self.device)), "vt_marker_detect")
        return markers, counts, binary

    def _select(self, cand: np.ndarray, filter_coords=None, filter_threshold=5) -> np.ndarray:
        """detect_markers :185-241 after the centroid list exists (host logic on a few dozen points, as in the reference)."""
        candidate_markers = cand
        if filter_coords is not None and len(candidate_markers) > 0:
            fc = np.array(filter_coords)
            keep = np.ones(len(candidate_markers), dtype=bool)
            for i, marker in enumerate(candidate_markers):
                distances = np.sqrt(np.sum((fc - marker) ** 2))
                if np.any(distances < filter_threshold):
                    keep[i] = False
            candidate_markers = candidate_markers[keep]
        expected = self.expected_markers
        if len(candidate_markers) == expected:
            return candidate_markers
        if len(candidate_markers) > expected:
            from sklearn.cluster import KMeans
            kmeans = KMeans(n_clusters=expected, random_state=0).fit(candidate_markers)
            refined = []
            for i in range(expected):
                pts = candidate_markers[kmeans.labels_ == i]
                if len(pts) > 0:
                    d = np.sqrt(np.sum((pts - kmeans.cluster_centers_[i]) ** 2, axis=1))
                    refined.append(pts[np.argmin(d)])
            return np.array(refined)
        if 0 < len(candidate_markers) < expected:
            print(f"Warning: Expected {expected} markers but found only {len(candidate_markers)}")
            return candidate_markers
        print("Warning: No markers detected")
        return np.array([])

    # ------------------------------------------------------------------ the reference's surface
    def calibrate(self, calibration_frame):
        markers = self.detect_markers(self.preprocess_frame(calibration_frame))
        self.baseline_markers = markers
        if self.grid_dims is None:
            n = len(markers)
            g = int(np.sqrt(n))
            self.grid_dims = (g, n // g)
            print(f"Estimated grid dimensions: {self.grid_dims}, detected {n} markers")
        self.ideal_grid = self.create_ideal_grid(markers)
        return markers

    def preprocess_frame(self, frame):
        """-> the binary `processed_frame` (uint8 0/255 numpy, like cv2) of one frame.  The reference's two-call protocol
        (`detect_markers(preprocess_frame(frame))`) is kept by remembering the centroids computed in the same device pass."""
        t = self._frames(frame)
        markers, counts, binary = self._detect(t, want_binary=True)
        n = int(counts[0])
        if n > markers.shape[1] or n > MAX_CAND:
            raise L.VtError(f"marker tracker: {n} candidate blobs overflow the device buffers")
        out = (binary[0] * 255).cpu().numpy()
        self._last = (out, markers[0, :n].cpu().numpy().astype(np.int64))
        return out

    def detect_markers(self, processed_frame, filter_coords=None, filter_threshold=5):
        """processed_frame: what `preprocess_frame` returned (its centroids were computed in the same device pass), or any
        binary image [H, W] (0 / non-zero), which is then labelled on the device from its pixels."""
        last = getattr(self, "_last", None)
        if last is not None and processed_frame is last[0]:
            cand = last[1]
        else:
            cand = self._centroids_of_binary(processed_frame)
        return self._select(cand, filter_coords, filter_threshold)

    def _centroids_of_binary(self, binary: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(binary)
        if b.ndim != 2:
            raise ValueError(f"processed_frame must be a 2-D binary image, got shape {b.shape}")
        t = torch.from_numpy((b != 0).astype(np.uint8))[None, :, :, None].to(self.device).contiguous()
        markers, counts, _ = self._detect(t, is_binary=True)
        n = int(counts[0])
        if n > markers.shape[1] or n > MAX_CAND:
            raise L.VtError(f"marker tracker: {n} candidate blobs overflow the device buffers")
        return markers[0, :n].cpu().numpy().astype(np.int64)

    def create_ideal_grid(self, markers):
        rows, cols = self.grid_dims
        x_min, y_min = np.min(markers, axis=0)
        x_max, y_max = np.max(markers, axis=0)
        x = np.linspace(x_min, x_max, cols)
        y = np.linspace(y_min, y_max, rows)
        return np.array([[j, i] for i in y for j in x])

    def get_marker_state(self, frame):
        current = self.detect_markers(self.preprocess_frame(frame))
        if self.baseline_markers is None:
            self.calibrate(frame)
            return np.zeros((len(current), 2))
        return self.match_and_compute_displacement(current)

    def match_and_compute_displacement(self, current_markers):
        if len(current_markers) == 0:
            return np.array([])
        cur = torch.from_numpy(np.ascontiguousarray(current_markers, dtype=np.int32)).to(self.device)[None]
        disp, _ = self._displacement(cur, torch.tensor([cur.shape[1]], dtype=torch.int32, device=self.device))
        return disp[0, :cur.shape[1]].cpu().numpy().astype(np.int64)

    def _displacement(self, markers: torch.Tensor, counts: torch.Tensor):
        N, mm = markers.shape[0], markers.shape[1]
        base = torch.from_numpy(np.ascontiguousarray(self.baseline_markers, dtype=np.int32)).to(self.device)
        disp = torch.zeros(N, mm, 2, dtype=torch.int32, device=self.device)
        force = torch.zeros(N, 3, dtype=torch.float64, device=self.device)
        L.check(L.lib().vt_marker_displacement(L.ptr(markers.contiguous()), L.ptr(counts), N, mm, L.ptr(base), base.shape[0], L.ptr(disp), L.ptr(force),
                                               L.stream_ptr(self.device)), "vt_marker_displacement")
        return disp, force

    def estimate_force(self, displacement):
        if len(displacement) == 0:
            return 0, np.array([0, 0])
        avg = np.mean(displacement, axis=0)
        mag = np.linalg.norm(avg)
        return mag, (avg / mag if mag > 0 else np.array([0, 0]))

    # ------------------------------------------------------------------ batched labelling of a GelSight stream
    def track_frames(self, frames, calibrate_on_first: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """frames [N, H, W, 3] uint8 (BGR) -> (displacement [N, n_markers, 2], force_magnitude [N], force_direction [N, 2]),
        n_markers = the largest marker count of the stream (rows beyond a frame's own count are zero; counts in `.last_counts`).
        Frame 0 becomes the baseline when none is set (process_image_sequence, :412-420).  One device pass for the whole
        stream; frames with MORE than `expected_markers` candidates go through the host KMeans branch individually."""
        t = self._frames(frames)
        markers, counts, _ = self._detect(t)
        cnt = counts.cpu().numpy()
        if (cnt > markers.shape[1]).any():
            raise L.VtError("marker tracker: candidate overflow")
        if self.baseline_markers is None and calibrate_on_first:
            self.baseline_markers = self._select(markers[0, :cnt[0]].cpu().numpy().astype(np.int64))
            self.ideal_grid = self.create_ideal_grid(self.baseline_markers)
        surplus = np.nonzero(cnt > self.expected_markers)[0]
        if len(surplus):
            mk = markers.cpu().numpy()
            for f in surplus:
                sel = self._select(mk[f, :cnt[f]].astype(np.int64))
                mk[f] = 0
                mk[f, :len(sel)] = sel
                cnt[f] = len(sel)
            markers = torch.from_numpy(mk).to(self.device)
            counts = torch.from_numpy(cnt.astype(np.int32)).to(self.device)
        disp, force = self._displacement(markers, counts)
        self.last_counts = cnt
        nmax = int(cnt.max()) if len(cnt) else 0
        f = force.cpu().numpy()
        return disp[:, :nmax].cpu().numpy().astype(np.int64), f[:, 0], f[:, 1:]
