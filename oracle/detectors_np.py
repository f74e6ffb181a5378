"""Oracle-side composition of the primitives into what the reference detectors compute.

TEST INFRASTRUCTURE ONLY.  ``edge_map`` restates ``ContentDetector._detect_edges``
(reference ``scenedetect/detectors/content_detector.py:213-239``) with the real ``numpy.median``
and the C restatement of cv2.Canny / cv2.dilate; ``score_batch`` adds the edge term to the integer
records of ``oracle.lib.score_batch``.  PARITY UNPINNED at the cv2 boundary (see oracle/cv2_restate.c).
"""

import math

import numpy as np

from oracle import lib as _orc


def estimated_kernel_size(frame_width: int, frame_height: int) -> int:
    size = 4 + round(math.sqrt(frame_width * frame_height) / 192)
    return size + 1 if size % 2 == 0 else size


def hsv_planes(frame: np.ndarray):
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w, _ = frame.shape
    hp, sp, vp = (np.empty((h, w), np.uint8) for _ in range(3))
    _orc.lib().orc_bgr2hsv_planes(frame.ctypes.data, w * 3, hp.ctypes.data, sp.ctypes.data, vp.ctypes.data, h, w)
    return hp, sp, vp


def canny_thresholds(lum: np.ndarray) -> tuple[int, int]:
    sigma = 1.0 / 3.0
    median = np.median(lum)
    low = int(max(0, (1.0 - sigma) * median))
    high = int(min(255, (1.0 + sigma) * median))
    return low, high


def edge_map(frame: np.ndarray, kernel_size: int = 0) -> np.ndarray:
    """Dilated Canny edge map (0/255) of the V plane of a BGR frame."""
    _, _, lum = hsv_planes(frame)
    h, w = lum.shape
    k = kernel_size or estimated_kernel_size(w, h)
    low, high = canny_thresholds(lum)
    edges = np.empty((h, w), np.uint8)
    _orc.lib().orc_canny(lum.ctypes.data, w, h, w, float(low), float(high), edges.ctypes.data)
    out = np.empty((h, w), np.uint8)
    _orc.lib().orc_dilate_rect(edges.ctypes.data, w, h, w, k, k, out.ctypes.data)
    return out


def score_batch(frames: np.ndarray, prev: np.ndarray | None = None, edges: bool = False, kernel_size: int = 0):
    """Per-frame records incl. ``edge_xor`` when ``edges`` is set."""
    rec = _orc.score_batch(frames, prev)
    if edges:
        last = edge_map(prev, kernel_size) if prev is not None else None
        for t in range(len(frames)):
            cur = edge_map(frames[t], kernel_size)
            if last is not None:
                rec["edge_xor"][t] = int(np.count_nonzero(cur != last))
            last = cur
    return rec


class OracleEngine:
    """Drop-in for ``pyscenedetect_amd.engine.ScoringEngine.score_host`` backed by the CPU oracle.

    Lets the host logic (detectors, SceneManager, sharding) be tested without a GPU.  Never used
    by the product."""

    def score_host(self, frames, prev=None, flags=7, edge_kernel=0, downscale=1.0, interpolation=1):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if downscale > 1.0:
            import cv2  # the shim

            def rs(f):
                return cv2.resize(f, (max(1, round(f.shape[1] / downscale)), max(1, round(f.shape[0] / downscale))),
                                  interpolation=interpolation)

            frames = np.stack([rs(f) for f in frames])
            prev = rs(prev) if prev is not None else None
        rec = score_batch(frames, prev, edges=bool(flags & 8), kernel_size=edge_kernel)
        if not flags & 1:
            rec["sad_h"] = rec["sad_s"] = rec["sad_v"] = 0
        if not flags & 6:
            rec["hist"] = 0
            rec["byte_sum"] = 0
        return rec

    def downscale_host(self, frames, downscale, interpolation=1):
        """The frames as the reference's detectors / callbacks receive them (scene_manager.py:666-678)."""
        import cv2  # the shim

        return np.stack([cv2.resize(f, (max(1, round(f.shape[1] / downscale)), max(1, round(f.shape[0] / downscale))),
                                    interpolation=interpolation) for f in np.ascontiguousarray(frames, dtype=np.uint8)])

    def hash_thumbs_host(self, frames, size, downscale=1.0, interpolation=1):
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if downscale > 1.0:
            import cv2  # the shim

            frames = np.stack([cv2.resize(f, (max(1, round(f.shape[1] / downscale)), max(1, round(f.shape[0] / downscale))),
                                          interpolation=interpolation) for f in frames])
        from oracle import lib as _lib

        return _lib.hash_thumbs(frames, size)
