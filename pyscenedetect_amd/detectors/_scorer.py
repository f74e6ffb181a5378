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
        self._seed = None           # a predecessor handed over by a SceneManager's shared pass, not yet on the device (seed())
        self.scored_since_seed = False

    @property
    def engine(self):
        if self._engine is None:
            from pyscenedetect_amd.engine import default_engine

            self._engine = default_engine()
        return self._engine

    def seed(self, frame: np.ndarray, scale=None) -> None:
        """``frame`` precedes whatever is scored next: a SceneManager's shared pass saw it on this detector's behalf (the reference's
        detector would hold the planes it derived from it, ``content_detector.py:189``), so a caller who goes on with
        ``process_frame()`` on the same detector gets the score against it.

        ``scale = (downscale factor, interpolation)`` the manager scored it behind.  The reference's detector holds the DOWNSCALED
        planes then, so the frames it accepts by hand afterwards are downscaled ones; the frame kept here is the one the stream
        delivered (making the small one at the end of every ``detect_scenes`` call would put a device round trip on the default
        pipeline), and :meth:`score` makes the small one the moment a frame of that size arrives -- a frame of the stream's own
        size is scored against the frame as delivered, as before (round 6; until then only the latter worked)."""
        # (the manager hands over its own copy of the frame, scene_manager.py: nothing else writes to it)
        self._last = self._seed = np.asarray(frame)
        self._last_shape = self._last.shape
        self._seed_scale = scale if scale is not None and scale[0] > 1.0 else None
        self.scored_since_seed = False

    def _seed_as_seen(self, shape) -> None:
        """A frame of ``shape`` follows a seed that was scored behind a downscale: if ``shape`` is the size the detectors saw, the seed
        becomes the frame they saw (``cv2.resize`` on the device, like every frame of that pass)."""
        scale = getattr(self, "_seed_scale", None)
        if self._seed is None or scale is None or tuple(shape) == tuple(self._seed.shape):
            return
        factor, interp = scale
        h, w = self._seed.shape[:2]
        if tuple(shape[:2]) == (max(1, round(h / factor)), max(1, round(w / factor))):
            self._last = self._seed = self.engine.downscale_host(self._seed[None], factor, interp)[0]
            self._last_shape = self._last.shape
            self._seed_scale = None

    def last_frame(self):
        """The frame scored last through this scorer (None before the first), for a SceneManager that takes over from direct
        ``process_frame()`` calls."""
        state = getattr(self, "_resident", None)
        if self._seed is None and state is not None and state["have_prev"]:
            h, w, _ = state["shape"]
            raw = state["bufs"][state["cur"] ^ 1].download(h * w * 3)      # (the buffer the last call filled)
            return raw.reshape(h, w, 3)
        return self._last

    def score(self, frame_img: np.ndarray, flags: int, edge_kernel: int = 0):
        frame = np.asarray(frame_img)
        if frame.dtype != np.uint8:
            raise ValueError("Image must be 8-bit BGR")
        if frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("Image must have three color channels")
        # A frame of another size than the one before: the reference's ContentDetector compares the planes it kept with the new
        # ones behind `assert len(left.shape) == 2 and left.shape == right.shape` (content_detector.py:29-36); the terms that
        # compare nothing across frames on the device (histogram, byte sum, thumbnails) go on, as in the reference.
        self._seed_as_seen(frame.shape)
        last_shape = getattr(self, "_last_shape", None)
        if last_shape is not None and last_shape != frame.shape and (flags & 9):      # HSV SAD | edges
            raise AssertionError("frame size changed from %dx%d to %dx%d" % (last_shape[1], last_shape[0], frame.shape[1], frame.shape[0]))
        self._last_shape = frame.shape
        self.scored_since_seed = True
        engine = self.engine
        if hasattr(engine, "alloc") and hasattr(engine, "score_device"):
            return self._score_resident(engine, frame, flags, edge_kernel)
        prev = self._last
        if prev is not None and prev.shape != frame.shape:
            prev = None
        rec = engine.score_host(frame[None], prev=prev, flags=flags, edge_kernel=edge_kernel)[0]
        self._last = np.array(frame, copy=True)
        self._seed = None
        return rec

    def _score_resident(self, engine, frame: np.ndarray, flags: int, edge_kernel: int):
        """Keep the previous frame in HBM (two ping-pong buffers): one upload per call instead of two."""
        h, w, _ = frame.shape
        nbytes = (h * w * 3 + 15) & ~15
        state = getattr(self, "_resident", None)
        if state is None or state["shape"] != frame.shape or state["engine"] is not engine:
            state = {"shape": frame.shape, "engine": engine, "bufs": [engine.alloc(nbytes), engine.alloc(nbytes)],
                     "cur": 0, "have_prev": False}
            self._resident = state
        cur = state["cur"]
        buf, other = state["bufs"][cur], state["bufs"][cur ^ 1]
        if self._seed is not None:              # a predecessor from a manager's pass: it goes where the previous frame would be
            if self._seed.shape == frame.shape:
                other.upload(np.ascontiguousarray(self._seed).reshape(-1))
                state["have_prev"] = True
            self._seed = None
            self._last = None                   # (the resident path keeps its frames on the device)
        buf.upload(np.ascontiguousarray(frame).reshape(-1))
        rec = engine.score_device(buf.ptr, 1, h, w, d_prev=other.ptr if state["have_prev"] else None, flags=flags,
                                  edge_kernel=edge_kernel)[0]
        state["cur"] = cur ^ 1
        state["have_prev"] = True
        return rec
