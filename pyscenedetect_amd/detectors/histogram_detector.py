"""HistogramDetector on the MI355X scoring engine
(reference ``scenedetect/detectors/histogram_detector.py:26-168``).

The device returns the exact 256-bin histogram of BT.601 luma (what ``cv2.cvtColor(BGR2YUV)`` +
``cv2.calcHist`` count); re-binning, ``cv2.normalize`` (L2, float32) and
``cv2.compareHist(HISTCMP_CORREL)`` (float64) are restated here operation for operation.
"""

import ctypes
import math
import sys
import typing as ty

import numpy as np

from pyscenedetect_amd import _native
from pyscenedetect_amd.detector import SceneDetector, plug_in_api
from pyscenedetect_amd.detectors._scorer import FrameScorer
from pyscenedetect_amd.timecode import FrameTimecode, give_back


def bin_lut(bins: int) -> np.ndarray:
    """cv2.calcHist's 8-bit lookup for ``bins`` uniform bins over [0,256): floor(j*bins/256)."""
    j = np.arange(256, dtype=np.float64)
    return np.clip(np.floor(j * (bins / 256.0)).astype(np.int64), 0, bins - 1)


def normalized_histogram(hist256: np.ndarray, bins: int) -> np.ndarray:
    """``cv2.normalize(cv2.calcHist([y],[0],None,[bins],[0,256])).flatten()`` from luma counts."""
    counts = np.zeros(bins, np.uint64)
    np.add.at(counts, bin_lut(bins), hist256.astype(np.uint64))
    h = counts.astype(np.float32)  # exact below 2**24, one rounding above (as OpenCV's int->f32)
    ss = float(np.cumsum(h.astype(np.float64) ** 2)[-1])  # double accumulation, exact for counts
    nrm = math.sqrt(ss)
    scale = np.float32(1.0 / nrm if nrm > sys.float_info.epsilon else 0.0)
    return h * scale  # float32 multiply, as convertTo(32F->32F, scale)


def _seq_sum(x: np.ndarray) -> float:
    """Left-to-right double sum (``np.sum`` is pairwise and would round differently)."""
    return float(np.cumsum(x)[-1]) if x.size else 0.0


def compare_hist_correl(h1: np.ndarray, h2: np.ndarray) -> float:
    """``cv2.compareHist(h1, h2, HISTCMP_CORREL)``: double sums in the order of OpenCV's 2-lane
    f64 SIMD loop (even/odd elements over the multiple-of-4 body, then the tail)."""
    a = h1.astype(np.float64)
    b = h2.astype(np.float64)
    n = a.size
    n4 = n - n % 4
    sums = []
    for x in (a, b, a * a, a * b, b * b):
        s = _seq_sum(x[0:n4:2]) + _seq_sum(x[1:n4:2])
        for j in range(n4, n):
            s += float(x[j])
        sums.append(s)
    s1, s2, s11, s12, s22 = sums
    scale = 1.0 / n
    num = s12 - s1 * s2 * scale
    denom2 = (s11 - s1 * s1 * scale) * (s22 - s2 * s2 * scale)
    return num / math.sqrt(denom2) if abs(denom2) > sys.float_info.epsilon else 1.0


def _native_normalized(hist256: np.ndarray, bins: int) -> np.ndarray:
    """`normalized_histogram` through ``psd_epilogue_hist_normalize`` (same arithmetic, ~20x less Python)."""
    h = np.ascontiguousarray(hist256, dtype=np.uint32)
    out = np.empty(bins, np.float32)
    _native.check(_native.load().psd_epilogue_hist_normalize(h.ctypes.data, int(bins), out.ctypes.data))
    return out


def _native_correl(h1: np.ndarray, h2: np.ndarray) -> float:
    """`compare_hist_correl` through ``psd_epilogue_hist_correl``."""
    out = ctypes.c_double(0.0)
    _native.check(_native.load().psd_epilogue_hist_correl(h1.ctypes.data, h2.ctypes.data, int(h1.size), ctypes.byref(out)))
    return out.value


class HistogramDetector(SceneDetector):
    METRIC_KEYS: ty.ClassVar[list[str]] = ["hist_diff"]

    def __init__(self, threshold: float = 0.20, bins: int = 128, min_scene_len=15, engine=None):
        super().__init__()
        # Internally the threshold is a correlation in [0, 1] (histogram_detector.py:50-52).
        self._threshold = max(0.0, min(1.0, 1.0 - threshold))
        self._bins = bins
        self._min_scene_len = min_scene_len
        self._last_hist = None
        self._last_cut = None
        self._metric_key = f"hist_diff [bins={self._bins}]"
        self._scorer = FrameScorer(engine)

    def get_metrics(self) -> list[str]:
        return [self._metric_key]

    def score_flags(self) -> int:
        return _native.SCORE_LUMA_HIST

    @staticmethod
    def calculate_histogram(frame_img: np.ndarray, bins: int = 256, normalize: bool = True, engine=None) -> np.ndarray:
        """Luma histogram of one BGR frame (reference ``histogram_detector.py:122-165``)."""
        rec = FrameScorer(engine).score(frame_img, _native.SCORE_LUMA_HIST)
        if normalize:
            return normalized_histogram(rec["hist"], bins)
        counts = np.zeros(bins, np.uint64)
        np.add.at(counts, bin_lut(bins), rec["hist"].astype(np.uint64))
        return counts.astype(np.float32).reshape(bins, 1)

    def process_record(self, timecode: FrameTimecode, record, height: int, width: int) -> list[FrameTimecode]:
        cut_list = []
        if not self._last_cut:
            self._last_cut = timecode
        hist = _native_normalized(record["hist"], self._bins)
        if self._last_hist is not None:
            hist_diff = _native_correl(self._last_hist, hist)
            if hist_diff <= self._threshold and ((timecode - self._last_cut) >= self._min_scene_len):
                cut_list.append(timecode)
                self._last_cut = timecode
            if self.stats_manager is not None:
                self.stats_manager.set_metrics(give_back(timecode), {self._metric_key: hist_diff})
        self._last_hist = hist
        return cut_list

    @plug_in_api
    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        if frame_img.dtype != np.uint8:
            raise ValueError("Image must be 8-bit rgb for HistogramDetector")
        if frame_img.shape[2] != 3:
            raise ValueError("Image must have three color channels for HistogramDetector")
        record = self._scorer.score(frame_img, self.score_flags())
        return self.process_record(timecode, record, frame_img.shape[0], frame_img.shape[1])
