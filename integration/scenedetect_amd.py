"""The binding a PySceneDetect maintainer would add, as a module that can be executed.

``install()`` patches the three seams of the UNMODIFIED reference package so that its own detectors, driven by its own
``SceneManager.detect_scenes``, take their pixel arithmetic from ``psd_score_batch`` (``include/psd_engine.h``):

  * ``ContentDetector._calculate_frame_score``  (``scenedetect/detectors/content_detector.py:147-190``; AdaptiveDetector
    inherits it): cv2.cvtColor + cv2.split + 3x ``_mean_pixel_distance`` + ``_detect_edges``  ->  PSD_SCORE_HSV_SAD | EDGES
  * ``HistogramDetector.calculate_histogram``     (``histogram_detector.py:122-165``): cv2.cvtColor(BGR2YUV) + cv2.split +
    cv2.calcHist -> PSD_SCORE_LUMA_HIST (cv2.normalize and cv2.compareHist stay the reference's calls)
  * ``numpy.mean(frame_img)`` in ``ThresholdDetector.process_frame`` (``threshold_detector.py:127``) -> PSD_SCORE_BYTE_SUM

Everything else -- weights, FlashFilter, the adaptive window, the fade state machine, StatsManager, timecodes --
stays the reference's code.  Only the C-ABI is touched: ``psd_create``, ``psd_score_batch``, ``psd_last_error``,
``psd_destroy`` and, for ContentDetector's frame-to-frame term, ``psd_device_alloc`` / ``psd_device_free`` /
``psd_memcpy_h2d`` / ``psd_score_batch_device`` (the previous frame stays in HBM: one upload per frame instead of two, and no
host copy of the frame -- the reference keeps three HSV planes per detector for the same purpose, content_detector.py:189)
on whatever shared library ``lib_path`` names (``pyscenedetect_amd/libpsd_hip.so`` on a machine with an MI355X).
Nothing of ``pyscenedetect_amd`` is imported.

``detect_many()`` is the batch front end of INTEGRATION.md B: what ``for video in dataset: detect(video, detector_cls())``
(``benchmark/__main__.py:44-61``) computes, for decoded videos of one size packed into ONE device batch --
``psd_score_segments_downscaled_device`` resizes every frame the way the reference's SceneManager does by default and scores it --
with the decisions still taken by the reference's own detector objects, fed from the records through the same three seams.
"""

import ctypes

import numpy as np

HSV_SAD, LUMA_HIST, BYTE_SUM, EDGES = 1, 2, 4, 8


class FrameScores(ctypes.Structure):            # psd_frame_scores, 1064 bytes
    _fields_ = [("sad_h", ctypes.c_uint64), ("sad_s", ctypes.c_uint64), ("sad_v", ctypes.c_uint64),
                ("edge_xor", ctypes.c_uint64), ("byte_sum", ctypes.c_uint64), ("hist", ctypes.c_uint32 * 256)]


class Binding:
    """One engine behind the C-ABI."""

    def __init__(self, lib_path: str, device: int = 0):
        lib = ctypes.CDLL(lib_path)
        lib.psd_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.psd_destroy.argtypes = [ctypes.c_void_p]
        lib.psd_destroy.restype = None
        lib.psd_score_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                        ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
        lib.psd_last_error.restype = ctypes.c_char_p
        lib.psd_device_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        lib.psd_device_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.psd_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        lib.psd_score_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                               ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.psd_hash_thumbs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t,
                                        ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        lib.psd_score_segments_downscaled_device.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self._lib = lib
        self._engine = ctypes.c_void_p()
        self._pairs = []
        self.current = None      # replay mode (detect_many): (record, has a predecessor) of the frame the seams are asked about
        if lib.psd_create(device, ctypes.byref(self._engine)) != 0:
            raise RuntimeError(lib.psd_last_error().decode())

    def close(self):
        if self._engine:
            for pair in self._pairs:
                pair.release()
            self._pairs = []
            self._lib.psd_destroy(self._engine)
            self._engine = ctypes.c_void_p()

    def _check(self, rc):
        if rc != 0:
            raise (ValueError if rc == -1 else MemoryError if rc == -5 else RuntimeError)(self._lib.psd_last_error().decode())

    def frame_pair(self) -> "FramePair":
        """State of ONE detector's frame-to-frame comparison: the frame it saw last stays in device memory."""
        pair = FramePair(self)
        self._pairs.append(pair)
        return pair

    def score(self, frame: np.ndarray, prev: np.ndarray | None, flags: int, kernel: int = 0) -> FrameScores:
        if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("Image must be 8-bit BGR")
        frame = np.ascontiguousarray(frame)
        if prev is not None:
            prev = np.ascontiguousarray(prev)
        rec = FrameScores()
        h, w, _ = frame.shape
        rc = self._lib.psd_score_batch(self._engine, frame.ctypes.data, 1, h, w, frame.strides[0], frame.nbytes,
                                       prev.ctypes.data if prev is not None else None, flags, kernel, ctypes.byref(rec))
        if rc != 0:
            raise (ValueError if rc == -1 else RuntimeError)(self._lib.psd_last_error().decode())
        return rec


    def score_videos_downscaled(self, videos, dst_h: int, dst_w: int, flags: int, kernel: int = 0):
        """Records of several decoded videos of ONE size (``uint8[n_i, H, W, 3]`` each), every frame resized to ``dst_w x dst_h`` with
        INTER_LINEAR first: the videos are packed back to back into one device buffer and scored by ONE
        ``psd_score_segments_downscaled_device`` call; returns one ``FrameScores`` array per video."""
        h, w = videos[0].shape[1:3]
        if any(v.dtype != np.uint8 or v.ndim != 4 or v.shape[1:] != (h, w, 3) for v in videos):
            raise ValueError("videos must be uint8[n, H, W, 3] of one size")
        stride = h * w * 3
        counts = [int(v.shape[0]) for v in videos]
        total = sum(counts)
        first = np.cumsum([0] + counts[:-1]).astype(np.int32)
        first = first[[c > 0 for c in counts]]                     # (a video without frames has no first frame)
        first = np.ascontiguousarray(np.unique(first))
        out = (FrameScores * max(total, 1))()
        if total == 0:
            return [out[0:0] for _ in videos]
        d = ctypes.c_void_p()
        self._check(self._lib.psd_device_alloc(self._engine, total * stride, ctypes.byref(d)))
        try:
            off = 0
            for v in videos:
                if v.shape[0]:
                    v = np.ascontiguousarray(v)
                    self._check(self._lib.psd_memcpy_h2d(self._engine, d.value + off * stride, v.ctypes.data, v.nbytes))
                    off += v.shape[0]
            self._check(self._lib.psd_score_segments_downscaled_device(
                self._engine, d, total, h, w, stride, first.ctypes.data, len(first), int(dst_h), int(dst_w), 1, flags, kernel,
                ctypes.byref(out), None))
        finally:
            self._lib.psd_device_free(self._engine, d)
        parts, off = [], 0
        for c in counts:
            parts.append(out[off:off + c])
            off += c
        return parts

    def hash_thumb(self, frame: np.ndarray, size: int) -> np.ndarray:
        """``cv2.resize(cv2.cvtColor(frame, BGR2GRAY), (size, size), INTER_AREA)`` from the device (``psd_hash_thumbs``)."""
        if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("Image must be 8-bit BGR")
        frame = np.ascontiguousarray(frame)
        h, w, _ = frame.shape
        out = np.empty((size, size), np.uint8)
        self._check(self._lib.psd_hash_thumbs(self._engine, frame.ctypes.data, 1, h, w, frame.strides[0], frame.nbytes, int(size),
                                              out.ctypes.data))
        return out


class FramePair:
    """Two device buffers that take turns holding "this frame" and "the frame before" of one detector."""

    def __init__(self, binding: Binding):
        self._b = binding
        self._bufs = [ctypes.c_void_p(), ctypes.c_void_p()]
        self._shape = None
        self._cur = 0
        self._have_prev = False

    def release(self):
        b = self._b
        for buf in self._bufs:
            if buf and b._engine:
                b._lib.psd_device_free(b._engine, buf)
        self._bufs = [ctypes.c_void_p(), ctypes.c_void_p()]
        self._shape, self._have_prev = None, False

    def score_next(self, frame: np.ndarray, flags: int, kernel: int = 0):
        """(record of ``frame`` against the frame of the previous call, whether there was one)."""
        if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3:
            raise ValueError("Image must be 8-bit BGR")
        frame = np.ascontiguousarray(frame)
        b = self._b
        if self._have_prev and self._shape != frame.shape:
            # the reference compares the planes it kept with the new ones behind `assert left.shape == right.shape`
            # (content_detector.py:29-36) and keeps the old planes: same error, same state
            raise AssertionError("frame size changed from %dx%d to %dx%d" % (self._shape[1], self._shape[0], frame.shape[1], frame.shape[0]))
        if self._shape != frame.shape:
            self.release()
            for i in range(2):
                b._check(b._lib.psd_device_alloc(b._engine, frame.nbytes, ctypes.byref(self._bufs[i])))
            self._shape = frame.shape
        cur, other = self._bufs[self._cur], self._bufs[self._cur ^ 1]
        b._check(b._lib.psd_memcpy_h2d(b._engine, cur, frame.ctypes.data, frame.nbytes))
        rec = FrameScores()
        h, w, _ = frame.shape
        had_prev = self._have_prev
        b._check(b._lib.psd_score_batch_device(b._engine, cur, 1, h, w, w * 3, frame.nbytes, other if had_prev else None, flags, kernel,
                                               ctypes.byref(rec), None))
        self._cur ^= 1
        self._have_prev = True
        return rec, had_prev


class _NumpyWithDeviceMean:
    """Stands in for the ``numpy`` name inside threshold_detector.py: ``mean`` of a BGR frame comes from the device."""

    def __init__(self, binding: Binding):
        self._b = binding

    def __getattr__(self, name):
        return getattr(np, name)

    def mean(self, a, *args, **kwargs):
        if not args and not kwargs and isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
            rec = self._b.current[0] if self._b.current is not None else self._b.score(a, None, BYTE_SUM)
            return np.float64(rec.byte_sum) / np.float64(a.size)        # integer-exact sum, one float64 divide
        return np.mean(a, *args, **kwargs)


def install(binding: Binding):
    """Patch the reference package (must be importable as ``scenedetect``).  Returns a function that undoes it."""
    import cv2
    from scenedetect.detectors import content_detector, histogram_detector, threshold_detector
    from scenedetect.detectors.content_detector import ContentDetector
    from scenedetect.detectors.hash_detector import HashDetector
    from scenedetect.detectors.histogram_detector import HistogramDetector

    saved = (ContentDetector._calculate_frame_score, HistogramDetector.calculate_histogram, threshold_detector.numpy,
             HashDetector.hash_frame)

    def _calculate_frame_score(self, timecode, frame_img):
        calculate_edges = (self._weights.delta_edges > 0.0) or self.stats_manager is not None
        pair = getattr(self, "_amd_pair", None)
        if pair is None and binding.current is None:
            pair = self._amd_pair = binding.frame_pair()        # this detector's previous frame lives in HBM
        kernel = int(self._kernel.shape[0]) if self._kernel is not None else 0
        if binding.current is not None:      # detect_many: the record was computed with the batch
            rec, had_prev = binding.current
        else:
            rec, had_prev = pair.score_next(frame_img, HSV_SAD | (EDGES if calculate_edges else 0), kernel)
        if not had_prev:
            return 0.0
        n = float(frame_img.shape[0] * frame_img.shape[1])
        # numpy scalars, as `_mean_pixel_distance` returns them (content_detector.py:29-36): the arithmetic below is then the
        # reference's own, weights that sum to zero included (NaN and a RuntimeWarning, no ZeroDivisionError)
        score_components = ContentDetector.Components(
            delta_hue=np.int64(rec.sad_h) / n, delta_sat=np.int64(rec.sad_s) / n, delta_lum=np.int64(rec.sad_v) / n,
            delta_edges=np.int64(255 * rec.edge_xor) / n if calculate_edges else 0.0)
        frame_score = sum(c * w for (c, w) in zip(score_components, self._weights, strict=True)) / sum(abs(w) for w in self._weights)
        if self.stats_manager is not None:
            metrics = {self.FRAME_SCORE_KEY: frame_score}
            metrics.update(score_components._asdict())
            self.stats_manager.set_metrics(timecode, metrics)
        return frame_score

    def calculate_histogram(frame_img, bins: int = 256, normalize: bool = True):
        rec = binding.current[0] if binding.current is not None else binding.score(frame_img, None, LUMA_HIST)
        counts = np.frombuffer(rec.hist, dtype=np.uint32)
        # cv2.calcHist with `bins` uniform bins over [0, 256): value v falls into bin floor(v * bins / 256)
        lut = np.floor(np.arange(256) * (bins / 256.0)).astype(np.int64)
        hist = np.zeros((bins, 1), np.float32)
        np.add.at(hist[:, 0], lut, counts.astype(np.float32))      # exact: counts < 2^24
        if normalize:
            hist = cv2.normalize(hist, hist).flatten()
        return hist

    def hash_frame(frame_img, hash_size, factor) -> np.ndarray:
        # the grey INTER_AREA thumbnail comes from the device (hash_detector.py:125-129); the lines below it stay as they are (:131-151)
        resized_img = binding.hash_thumb(frame_img, hash_size * factor)
        max_value = np.max(np.max(resized_img))
        if max_value == 0:
            max_value = 1
        resized_img = np.asarray(np.float32(resized_img) / max_value)
        dct_low_freq = cv2.dct(resized_img)[:hash_size, :hash_size]
        med = np.median(np.asarray(dct_low_freq, dtype=np.float32))
        return dct_low_freq > med

    ContentDetector._calculate_frame_score = _calculate_frame_score
    HashDetector.hash_frame = staticmethod(hash_frame)
    HistogramDetector.calculate_histogram = staticmethod(calculate_histogram)
    threshold_detector.numpy = _NumpyWithDeviceMean(binding)

    def uninstall():
        ContentDetector._calculate_frame_score = saved[0]
        HistogramDetector.calculate_histogram = staticmethod(saved[1])
        threshold_detector.numpy = saved[2]
        HashDetector.hash_frame = staticmethod(saved[3])
        _ = content_detector, histogram_detector

    return uninstall


def detect_many(binding: Binding, videos, make_detector, fps=25.0):
    """Cut lists of several decoded videos of one size, computed as the reference's benchmark computes them one by one --
    ``detect(video, detector_cls())`` (``benchmark/__main__.py:44-61``): a default SceneManager, i.e. every frame resized by
    ``compute_downscale_factor(max(frame_size))`` (``scene_manager.py:123-140,525-528,666-678``), no StatsManager -- from ONE device batch.
    ``install(binding)`` must be in effect.  The pixel work is one ``psd_score_segments_downscaled_device`` call; every decision is
    taken by a fresh detector of the reference (``make_detector()``) fed frame by frame through ``process_frame`` / ``post_process``,
    whose seams answer from the records.  Returns ``[[cut frame numbers], ...]`` as ``SceneManager.get_cut_list`` orders them."""
    from scenedetect import FrameTimecode
    from scenedetect.detectors import AdaptiveDetector, ContentDetector, HistogramDetector, ThresholdDetector
    from scenedetect.scene_manager import compute_downscale_factor

    h, w = videos[0].shape[1:3]
    factor = compute_downscale_factor(max(w, h))
    dst_h, dst_w = (max(1, round(h / factor)), max(1, round(w / factor))) if factor > 1.0 else (h, w)
    probe = make_detector()
    if isinstance(probe, (ContentDetector, AdaptiveDetector)):
        flags = HSV_SAD | (EDGES if probe._weights.delta_edges > 0.0 else 0)
        kernel = int(probe._kernel.shape[0]) if probe._kernel is not None else 0
    elif isinstance(probe, HistogramDetector):
        flags, kernel = LUMA_HIST, 0
    elif isinstance(probe, ThresholdDetector):
        flags, kernel = BYTE_SUM, 0
    else:
        raise TypeError("detect_many drives Content / Adaptive / Histogram / Threshold detectors")
    records = binding.score_videos_downscaled(videos, dst_h, dst_w, flags, kernel)
    seen = np.empty((dst_h, dst_w, 3), np.uint8)       # what a detector is handed: only its shape and dtype are looked at (the seams answer)
    out = []
    try:
        for video, recs in zip(videos, records):
            det = make_detector()
            cuts, tc = [], None
            for t in range(video.shape[0]):
                binding.current = (recs[t], t > 0)
                tc = FrameTimecode(t, fps)
                cuts += det.process_frame(tc, seen)
            if tc is not None:
                cuts += det.post_process(tc)
            out.append(sorted({int(c.frame_num) for c in cuts}))
    finally:
        binding.current = None
    return out
