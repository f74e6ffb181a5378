"""The reference's execution model for the hot path, timed as the CPU baseline -- TEST / BENCH INFRASTRUCTURE ONLY.

``/root/reference`` does not exist on the GPU box, so ``bench.py`` cannot import the reference there.  What the
reference executes per frame on this path is restated here operation by operation in the same libraries
(numpy for what it does in numpy, the cv2 shim over ``oracle/cv2_restate.c`` for what it does in OpenCV), one
Python process, one thread, frame by frame -- ``SceneManager._process_frame`` -> ``ContentDetector.process_frame``
(reference ``scenedetect/scene_manager.py:578-585``, ``scenedetect/detectors/content_detector.py:29-36,147-190``).
The numpy half is the reference's own expression; the OpenCV half is the restatement (real OpenCV is not installed).
"""

import time

import numpy as np


def mean_pixel_distance(left: np.ndarray, right: np.ndarray) -> float:
    """content_detector.py:29-36, verbatim arithmetic."""
    num_pixels = float(left.shape[0] * left.shape[1])
    return np.sum(np.abs(left.astype(np.int32) - right.astype(np.int32))) / num_pixels


def content_scores_loop(frames: np.ndarray) -> np.ndarray:
    """content_val per frame as ContentDetector._calculate_frame_score computes it with the default weights."""
    import cv2  # the oracle shim (oracle/cv2_shim)

    weights = (1.0, 1.0, 1.0, 0.0)
    out = np.zeros(len(frames))
    last = None
    for i, frame in enumerate(frames):
        hue, sat, lum = cv2.split(cv2.cvtColor(frame, cv2.COLOR_BGR2HSV))
        if last is not None:
            comps = (mean_pixel_distance(hue, last[0]), mean_pixel_distance(sat, last[1]), mean_pixel_distance(lum, last[2]), 0.0)
            out[i] = sum(c * w for c, w in zip(comps, weights)) / sum(abs(w) for w in weights)
        last = (hue, sat, lum)
    return out


def time_models(sample: np.ndarray) -> dict:
    """frames/s of (i) the per-frame Python loop above, one process, and (iii) its numpy-only half."""
    n = len(sample)
    t0 = time.perf_counter()
    scores = content_scores_loop(sample)
    t_loop = time.perf_counter() - t0
    planes = [np.ascontiguousarray(sample[i][:, :, c]) for i in range(min(n, 4)) for c in range(3)]
    t0 = time.perf_counter()
    reps = 0
    for i in range(3, len(planes)):
        mean_pixel_distance(planes[i], planes[i - 3])
        reps += 1
    t_np = (time.perf_counter() - t0) / max(reps, 1) * 3      # three planes per frame
    return {"python_loop_frames_per_s": round((n - 1) / t_loop, 2) if t_loop > 0 else None,
            "numpy_half_frames_per_s": round(1.0 / t_np, 2) if t_np > 0 else None,
            "python_loop_frames": n, "_scores": scores}
