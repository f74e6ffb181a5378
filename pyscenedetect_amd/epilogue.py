"""Whole-clip decisions from per-frame records, in native code (``psd_epilogue_*``).

The detectors in :mod:`pyscenedetect_amd.detectors` decide frame by frame in Python, like the
reference.  For throughput (hundreds of thousands of frames per second per GPU) the same logic is
also available over a whole clip's records at once; both produce identical cut lists and metrics
(tests/test_host_golden.py).  No GPU is involved here.
"""

import ctypes
from fractions import Fraction

import numpy as np

from pyscenedetect_amd import _native
from pyscenedetect_amd._native import AdaptiveParams, ContentParams, HashParams, HistParams, ThresholdParams
from pyscenedetect_amd.timecode import framerate_to_fraction, parse_timecode_seconds


def _fps(fps) -> Fraction:
    return fps if isinstance(fps, Fraction) else framerate_to_fraction(float(fps))


def _min_len(length, rate: Fraction, int_is_frames: bool = True) -> tuple[int, float]:
    """(min_len_frames, min_len_secs) for the C structs; secs < 0 means "frames"."""
    if isinstance(length, bool):
        raise TypeError("min_scene_len must be int, float or str")
    if isinstance(length, int):
        return int(length), -1.0
    if isinstance(length, float):
        return 0, float(length)
    if isinstance(length, str):
        s = length.strip()
        if s.isdigit():
            # FlashFilter treats a digit string as frames; FrameTimecode comparisons go through seconds.
            return (int(s), -1.0) if int_is_frames else (0, int(s) / float(rate))
        return 0, float(parse_timecode_seconds(s, rate))
    if hasattr(length, "seconds") and not isinstance(length, (bytes, bytearray)):   # FrameTimecode / Timecode: a duration
        return 0, float(length.seconds)
    raise TypeError(f"unsupported min_scene_len type {type(length)}")


def _recs(records: np.ndarray) -> np.ndarray:
    records = np.ascontiguousarray(records)
    if records.dtype != _native.RECORD_DTYPE:
        raise ValueError("records must have dtype RECORD_DTYPE")
    return records


def _sums(records: np.ndarray) -> tuple[np.ndarray, int]:
    """(array, stride in bytes) for the epilogues that only read the five sums: full records (``RECORD_DTYPE``) are read in
    place, records without the histogram (``SUMS_DTYPE``, ``ScoringEngine.collect(sums_only=True)``) likewise."""
    records = np.asarray(records)
    if records.dtype != _native.RECORD_DTYPE and records.dtype != _native.SUMS_DTYPE and records.dtype != _native.SUMS_DIFF_DTYPE:
        raise ValueError("records must have dtype RECORD_DTYPE, SUMS_DTYPE or SUMS_DIFF_DTYPE")
    if records.ndim != 1 or (len(records) > 1 and records.strides[0] < records.dtype.itemsize):
        records = np.ascontiguousarray(records).reshape(-1)
    return records, (records.strides[0] if len(records) > 1 else records.dtype.itemsize)


def content_scores(records, height: int, width: int, weights=(1.0, 1.0, 1.0, 0.0), first_has_prev: bool = False):
    """dict of float64 arrays: content_val, delta_hue, delta_sat, delta_lum, delta_edges."""
    records, stride = _sums(records)
    n = len(records)
    out = {k: np.zeros(n, np.float64) for k in ("content_val", "delta_hue", "delta_sat", "delta_lum", "delta_edges")}
    w = (ctypes.c_double * 4)(*[float(x) for x in weights])
    _native.check(_native.load().psd_epilogue_content_scores_sums(
        records.ctypes.data, stride, n, height, width, ctypes.cast(w, ctypes.c_void_p), int(first_has_prev),
        *[out[k].ctypes.data for k in ("content_val", "delta_hue", "delta_sat", "delta_lum", "delta_edges")]))
    return out


def content_cuts(content_val, fps, threshold: float = 27.0, min_scene_len=15, filter_mode: int = 0,
                 first_frame: int = 0) -> list[int]:
    rate = _fps(fps)
    cv = np.ascontiguousarray(content_val, np.float64)
    p = ContentParams()
    p.threshold = float(threshold)
    p.filter_mode = int(filter_mode)
    # A float-backed FrameTimecode length is seconds as well.
    p.min_len_frames, p.min_len_secs = _min_len(min_scene_len, rate, int_is_frames=True)
    cuts = np.zeros(len(cv) + 1, np.int64)
    nc = ctypes.c_int(0)
    _native.check(_native.load().psd_epilogue_content_cuts(cv.ctypes.data, len(cv), first_frame, rate.numerator,
                                                          rate.denominator, ctypes.byref(p), cuts.ctypes.data,
                                                          ctypes.byref(nc)))
    return cuts[: nc.value].tolist()


def adaptive_cuts(content_val, fps, adaptive_threshold: float = 3.0, min_scene_len=15, window_width: int = 2,
                  min_content_val: float = 15.0, first_frame: int = 0):
    """(cuts, adaptive_ratio[n] with NaN where the reference writes no metric)."""
    rate = _fps(fps)
    cv = np.ascontiguousarray(content_val, np.float64)
    p = AdaptiveParams()
    p.adaptive_threshold = float(adaptive_threshold)
    p.min_content_val = float(min_content_val)
    p.window_width = int(window_width)
    p.min_len_frames, p.min_len_secs = _min_len(min_scene_len, rate, int_is_frames=False)
    ratio = np.zeros(len(cv), np.float64)
    cuts = np.zeros(len(cv) + 1, np.int64)
    nc = ctypes.c_int(0)
    _native.check(_native.load().psd_epilogue_adaptive_cuts(cv.ctypes.data, len(cv), first_frame, rate.numerator,
                                                           rate.denominator, ctypes.byref(p), ratio.ctypes.data,
                                                           cuts.ctypes.data, ctypes.byref(nc)))
    return cuts[: nc.value].tolist(), ratio


def hist_cuts(records, fps, threshold: float = 0.20, bins: int = 128, min_scene_len=15, first_frame: int = 0):
    """(cuts, hist_diff[n] with NaN for the first frame)."""
    rate = _fps(fps)
    records = _recs(records)
    p = HistParams()
    p.threshold = float(threshold)
    p.bins = int(bins)
    p.min_len_frames, p.min_len_secs = _min_len(min_scene_len, rate, int_is_frames=False)
    diff = np.zeros(len(records), np.float64)
    cuts = np.zeros(len(records) + 1, np.int64)
    nc = ctypes.c_int(0)
    _native.check(_native.load().psd_epilogue_hist_cuts(records.ctypes.data, len(records), None, first_frame,
                                                       rate.numerator, rate.denominator, ctypes.byref(p),
                                                       diff.ctypes.data, cuts.ctypes.data, ctypes.byref(nc)))
    return cuts[: nc.value].tolist(), diff


def hist_cuts_from_diff(hist_diff, fps, threshold: float = 0.20, min_scene_len=15, first_frame: int = 0):
    """Cuts from ``hist_diff`` values computed elsewhere (``ScoringEngine.hist_diff_device``: NaN where a frame has no predecessor) -- the
    decision loop of :func:`hist_cuts`."""
    rate = _fps(fps)
    diff = np.ascontiguousarray(hist_diff, dtype=np.float64)
    p = HistParams()
    p.threshold = float(threshold)
    p.bins = 0
    p.min_len_frames, p.min_len_secs = _min_len(min_scene_len, rate, int_is_frames=False)
    cuts = np.zeros(len(diff) + 1, np.int64)
    nc = ctypes.c_int(0)
    _native.check(_native.load().psd_epilogue_hist_cuts_from_diff(diff.ctypes.data if len(diff) else None, len(diff), first_frame, rate.numerator,
                                                                 rate.denominator, ctypes.byref(p), cuts.ctypes.data, ctypes.byref(nc)))
    return cuts[: nc.value].tolist()


def threshold_cuts(records, height: int, width: int, fps, threshold: float = 12, min_scene_len=15,
                   fade_bias: float = 0.0, add_final_scene: bool = False, method: int = 0, first_frame: int = 0):
    """(cuts, average_rgb[n])."""
    rate = _fps(fps)
    records, stride = _sums(records)
    p = ThresholdParams()
    p.threshold = int(threshold)
    p.method = int(method)
    p.fade_bias = float(fade_bias)
    p.add_final_scene = int(bool(add_final_scene))
    p.min_len_frames, p.min_len_secs = _min_len(min_scene_len, rate, int_is_frames=False)
    avg = np.zeros(len(records), np.float64)
    cuts = np.zeros(len(records) + 1, np.int64)
    nc = ctypes.c_int(0)
    _native.check(_native.load().psd_epilogue_threshold_cuts_sums(records.ctypes.data, stride, len(records), height, width,
                                                                 first_frame, rate.numerator, rate.denominator,
                                                            ctypes.byref(p), avg.ctypes.data, cuts.ctypes.data,
                                                            ctypes.byref(nc)))
    return cuts[: nc.value].tolist(), avg


def hash_bits(thumbs, hash_size: int = 8) -> np.ndarray:
    """Perceptual hash bits bool[n, hash_size, hash_size] of grey thumbnails uint8[n, S, S] -- the back half of
    ``HashDetector.hash_frame`` (reference ``hash_detector.py:131-151``)."""
    thumbs = np.ascontiguousarray(thumbs, dtype=np.uint8)
    if thumbs.ndim != 3 or thumbs.shape[1] != thumbs.shape[2]:
        raise ValueError("thumbs must be uint8[n, S, S]")
    n, size, _ = thumbs.shape
    bits = np.zeros((n, hash_size, hash_size), np.uint8)
    _native.check(_native.load().psd_epilogue_hash_bits(thumbs.ctypes.data if n else None, n, size, int(hash_size),
                                                       bits.ctypes.data if n else None))
    return bits.astype(bool)


def hash_cuts(bits, fps, threshold: float = 0.35, min_scene_len=15, first_frame: int = 0):
    """(cuts, hash_dist[n] with NaN for the first frame) from hash bits bool[n, size, size]."""
    rate = _fps(fps)
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    n = len(bits)
    p = HashParams()
    p.threshold = float(threshold)
    p.hash_size = int(bits.shape[1]) if bits.ndim == 3 else 1
    p.min_len_frames, p.min_len_secs = _min_len(min_scene_len, rate, int_is_frames=False)
    dist = np.zeros(n, np.float64)
    cuts = np.zeros(n + 1, np.int64)
    nc = ctypes.c_int(0)
    _native.check(_native.load().psd_epilogue_hash_cuts(bits.ctypes.data if n else None, n, None, first_frame,
                                                       rate.numerator, rate.denominator, ctypes.byref(p),
                                                       dist.ctypes.data, cuts.ctypes.data, ctypes.byref(nc)))
    return cuts[: nc.value].tolist(), dist
