"""AdaptiveDetector: ContentDetector scores through a rolling-window ratio
(reference ``scenedetect/detectors/adaptive_detector.py:28-143``)."""

import numpy as np

from pyscenedetect_amd.detectors.content_detector import ContentDetector
from pyscenedetect_amd.timecode import FrameTimecode


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
        super().process_record(timecode, record, height, width)
        if self._frame_score is None:
            return []
        if self._last_cut is None:
            self._last_cut = timecode
        need = 1 + 2 * self.window_width
        self._buffer.append((timecode, self._frame_score))
        if len(self._buffer) < need:
            return []
        self._buffer = self._buffer[-need:]
        target_timecode, target_score = self._buffer[self.window_width]
        average = sum(score for i, (_tc, score) in enumerate(self._buffer) if i != self.window_width) / (
            2.0 * self.window_width
        )
        average_is_zero = abs(average) < 0.00001
        adaptive_ratio = 0.0
        if not average_is_zero:
            adaptive_ratio = min(target_score / average, 255.0)
        elif target_score >= self.min_content_val:
            adaptive_ratio = 255.0
        if self.stats_manager is not None:
            self.stats_manager.set_metrics(target_timecode, {self._adaptive_ratio_key: adaptive_ratio})
        threshold_met = adaptive_ratio >= self.adaptive_threshold and target_score >= self.min_content_val
        # Note: the *current* position is compared with the last cut, the *target* is emitted
        # (adaptive_detector.py:139-142).
        min_length_met = (timecode - self._last_cut) >= self.min_scene_len
        if threshold_met and min_length_met:
            self._last_cut = target_timecode
            return [target_timecode]
        return []

    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        record = self._scorer.score(frame_img, self.score_flags(), self.edge_kernel_size())
        return self.process_record(timecode, record, frame_img.shape[0], frame_img.shape[1])
