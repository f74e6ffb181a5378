"""ContentDetector on the MI355X scoring engine.

Same constructor, metric keys and decisions as the reference
(``scenedetect/detectors/content_detector.py:49-243``).  The cv2/numpy pixel work of
``_calculate_frame_score`` (:147-190) -- BGR->HSV, three ``_mean_pixel_distance`` calls and the
optional Canny/dilate edge delta -- happens on the device and arrives here as exact integer sums;
this class turns them into ``content_val`` with the reference's float operations, in its order.
"""

import math
import operator
import typing as ty

import numpy as np

from pyscenedetect_amd import _native
from pyscenedetect_amd.detector import FlashFilter, SceneDetector, plug_in_api
from pyscenedetect_amd.detectors._scorer import FrameScorer
from pyscenedetect_amd.timecode import FrameTimecode, give_back


def estimated_kernel_size(frame_width: int, frame_height: int) -> int:
    """Edge-dilation size for a resolution (reference ``content_detector.py:39-46``)."""
    size = 4 + round(math.sqrt(frame_width * frame_height) / 192)
    return size + 1 if size % 2 == 0 else size


class ContentDetector(SceneDetector):
    """Detects fast cuts from the HSV (and optionally edge) change between adjacent frames."""

    class Components(ty.NamedTuple):
        delta_hue: float = 1.0
        delta_sat: float = 1.0
        delta_lum: float = 1.0
        delta_edges: float = 0.0

    DEFAULT_COMPONENT_WEIGHTS = Components()
    LUMA_ONLY_WEIGHTS = Components(delta_hue=0.0, delta_sat=0.0, delta_lum=1.0, delta_edges=0.0)
    FRAME_SCORE_KEY = "content_val"
    METRIC_KEYS: ty.ClassVar[list[str]] = [FRAME_SCORE_KEY, *Components._fields]

    def __init__(
        self,
        threshold: float = 27.0,
        min_scene_len=15,
        weights: "ContentDetector.Components" = DEFAULT_COMPONENT_WEIGHTS,
        luma_only: bool = False,
        kernel_size: int | None = None,
        filter_mode: FlashFilter.Mode = FlashFilter.Mode.MERGE,
        engine=None,
    ):
        super().__init__()
        self._threshold = threshold
        self._weights = ContentDetector.LUMA_ONLY_WEIGHTS if luma_only else weights
        self._kernel_size = None
        if kernel_size is not None:
            if kernel_size < 3 or kernel_size % 2 == 0:
                raise ValueError("kernel_size must be odd integer >= 3")
            # (the reference builds numpy.ones((kernel_size, kernel_size)) here, content_detector.py:137: an odd float such as
            #  27.0 gets past the range check and is a TypeError there)
            self._kernel_size = operator.index(kernel_size)
        self._frame_score: float | None = None
        self._flash_filter = FlashFilter(mode=filter_mode, length=min_scene_len)
        self._have_last = False
        self._scorer = FrameScorer(engine)

    def get_metrics(self):
        return ContentDetector.METRIC_KEYS

    @property
    def event_buffer_length(self) -> int:
        return self._flash_filter.max_behind

    # -- device terms ------------------------------------------------------------------------------
    def score_flags(self) -> int:
        flags = _native.SCORE_HSV_SAD
        # The reference computes edges whenever a StatsManager is attached (content_detector.py:158).
        if self._weights.delta_edges > 0.0 or self.stats_manager is not None:
            flags |= _native.SCORE_EDGES
        return flags

    def edge_kernel_size(self) -> int:
        return self._kernel_size or 0

    # -- decisions -------------------------------------------------------------------------------
    def _score_from_record(self, timecode: FrameTimecode, record, height: int, width: int) -> float:
        if not self._have_last:
            self._have_last = True
            return 0.0
        num_pixels = float(height * width)
        components = ContentDetector.Components(
            delta_hue=int(record["sad_h"]) / num_pixels,
            delta_sat=int(record["sad_s"]) / num_pixels,
            delta_lum=int(record["sad_v"]) / num_pixels,
            delta_edges=(255 * int(record["edge_xor"])) / num_pixels if self.score_flags() & _native.SCORE_EDGES else 0.0,
        )
        weighted = sum(c * w for c, w in zip(components, self._weights, strict=True))
        weight_sum = sum(abs(w) for w in self._weights)
        # The reference computes all of this on numpy scalars (`numpy.sum(...) / float(num_pixels)`, content_detector.py:29-36): the same
        # IEEE arithmetic as Python's floats -- except that weights which sum to zero give a NaN score and a RuntimeWarning there,
        # not a ZeroDivisionError -- and what it stores are numpy.float64 objects.  Python floats here (a third of the per-frame cost of
        # this function less), numpy where it shows.
        frame_score = weighted / weight_sum if weight_sum != 0.0 else np.float64(weighted) / weight_sum
        if self.stats_manager is not None:
            metrics = {self.FRAME_SCORE_KEY: np.float64(frame_score)}
            metrics.update({key: np.float64(value) for key, value in components._asdict().items()})
            self.stats_manager.set_metrics(give_back(timecode), metrics)
        return frame_score

    def process_record(self, timecode: FrameTimecode, record, height: int, width: int) -> list[FrameTimecode]:
        self._frame_score = self._score_from_record(timecode, record, height, width)
        above = self._frame_score >= self._threshold
        return self._flash_filter.filter(timecode=timecode, above_threshold=above)

    @plug_in_api
    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        record = self._scorer.score(frame_img, self.score_flags(), self.edge_kernel_size())
        return self.process_record(timecode, record, frame_img.shape[0], frame_img.shape[1])
