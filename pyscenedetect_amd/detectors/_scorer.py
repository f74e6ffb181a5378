"""Per-frame scoring helper for the one-frame-at-a-time ``process_frame()`` API."""

import numpy as np


class FrameScorer:
    """Scores one frame against the previous one on the device.

    Keeps a private copy of the last frame (the reference keeps the derived HSV planes,
    ``content_detector.py:189``) and sends both through ``psd_score_batch`` as a batch of one with
    a halo.  ``engine`` is any object with ``score_host(frames, prev, flags, edge_kernel)``;
    the default is the process-wide HIP engine, which raises if no GPU is present.
    """

    def __init__(self, engine=None):
        self._engine = engine
        self._last = None

    @property
    def engine(self):
        if self._engine is None:
            from pyscenedetect_amd.engine import default_engine

            self._engine = default_engine()
        return self._engine

    def score(self, frame_img: np.ndarray, flags: int, edge_kernel: int = 0):
        frame = np.asarray(frame_img)
        if frame.dtype != np.uint8:
            raise ValueError("Image must be 8-bit BGR")
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("Image must have three color channels")
        prev = self._last
        if prev is not None and prev.shape != frame.shape:
            prev = None
        rec = self.engine.score_host(frame[None], prev=prev, flags=flags, edge_kernel=edge_kernel)[0]
        self._last = np.array(frame, copy=True)
        return rec
