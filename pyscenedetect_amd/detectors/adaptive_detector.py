"""AdaptiveDetector: ContentDetector scores through a rolling-window ratio
(reference ``scenedetect/detectors/adaptive_detector.py:28-143``)."""

import numpy as np

from pyscenedetect_amd.detector import plug_in_api
from pyscenedetect_amd.detectors.content_detector import ContentDetector
from pyscenedetect_amd.timecode import FrameTimecode, give_back


class AdaptiveDetector(ContentDetector):
    ADAPTIVE_RATIO_KEY_TEMPLATE = "adaptive_ratio{luma_only} (w={window_width})"

    def __init__(
        self,
        adaptive_threshold: float = 3.0,
        min_scene_len=15,
        window_width: int = 2,
        min_content_val: float = 15.0,
        weights: ContentDetector.Components = ContentDetector.DEFAULT_COMPONENT_WEIGHTS,
        luma_only: bool = False,
        kernel_size: int | None = None,
        engine=None,
    ):
        if window_width < 1:
            raise ValueError("window_width must be at least 1.")
        # The parent never cuts on its own: threshold 255, no flash filter (adaptive_detector.py:71-77).
        super().__init__(threshold=255.0, min_scene_len=0, weights=weights, luma_only=luma_only,
                         kernel_size=kernel_size, engine=engine)
        self.min_scene_len = min_scene_len
        self.adaptive_threshold = adaptive_threshold
        self.min_content_val = min_content_val
        self.window_width = window_width
        self._adaptive_ratio_key = AdaptiveDetector.ADAPTIVE_RATIO_KEY_TEMPLATE.format(
            window_width=window_width, luma_only="" if not luma_only else "_lum"
        )
        self._buffer: list[tuple[FrameTimecode, float]] = []
        self._last_cut: FrameTimecode | None = None

    @property
    def event_buffer_length(self) -> int:
        return self.window_width

    def get_metrics(self) -> list[str]:
        return [*super().get_metrics(), self._adaptive_ratio_key]

    def process_record(self, timecode: FrameTimecode, record, height: int, width: int) -> list[FrameTimecode]:
        """Decide about the frame ``window_width`` frames back, now that its right-hand neighbours are known."""
        super().process_record(timecode, record, height, width)   # updates self._frame_score / the stats
        score_now = self._frame_score
        if score_now is None:
            return []
        if self._last_cut is None:
            self._last_cut = timecode            # scene length is measured from the first frame seen
        w = self.window_width
        window = self._buffer
        window.append((timecode, score_now))
        if len(window) < 2 * w + 1:
            return []
        del window[: len(window) - (2 * w + 1)]  # keep exactly w neighbours on each side of the candidate
        candidate_tc, candidate = window[w]
        # Mean of the 2w neighbours; accumulated left to right from 0 so that the float result is the
        # reference's sum(...) / (2.0 * w) bit for bit.
        neighbours = 0
        for pos, (_tc, value) in enumerate(window):
            if pos != w:
                neighbours = neighbours + value
        mean = neighbours / (2.0 * w)
        if abs(mean) < 0.00001:
            ratio = stored = 255.0 if candidate >= self.min_content_val else 0.0
        else:
            quotient = candidate / mean
            if 255.0 < quotient:             # min(quotient, 255.0): the cap is a Python float in the reference ...
                ratio = stored = 255.0
            else:                            # ... the quotient a numpy.float64 (its scores are), NaN included
                ratio = quotient
                stored = np.float64(quotient) if self.stats_manager is not None else quotient
        stats = self.stats_manager
        if stats is not None:
            stats.set_metrics(give_back(candidate_tc), {self._adaptive_ratio_key: stored})
        is_peak = ratio >= self.adaptive_threshold and candidate >= self.min_content_val
        # The reference measures the gap from the CURRENT position but reports the candidate
        # (adaptive_detector.py:139-142); kept as is.
        long_enough = (timecode - self._last_cut) >= self.min_scene_len
        if not (is_peak and long_enough):
            return []
        self._last_cut = candidate_tc
        return [candidate_tc]

    @plug_in_api
    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        record = self._scorer.score(frame_img, self.score_flags(), self.edge_kernel_size())
        return self.process_record(timecode, record, frame_img.shape[0], frame_img.shape[1])
