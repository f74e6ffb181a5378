"""ThresholdDetector (fade in/out) on the MI355X scoring engine
(reference ``scenedetect/detectors/threshold_detector.py:30-191``).

``numpy.mean(frame_img)`` (:127) is an exact integer sum followed by one float64 division; the
device supplies the sum (``byte_sum``).
"""

import warnings
from enum import Enum

import numpy as np

from pyscenedetect_amd import _native
from pyscenedetect_amd.detector import SceneDetector, plug_in_api
from pyscenedetect_amd.detectors._scorer import FrameScorer
from pyscenedetect_amd.timecode import FrameTimecode, give_back, new_like


class ThresholdDetector(SceneDetector):
    class Method(Enum):
        FLOOR = 0
        CEILING = 1

    THRESHOLD_VALUE_KEY = "average_rgb"

    def __init__(self, threshold: float = 12, min_scene_len=15, fade_bias: float = 0.0,
                 add_final_scene: bool = False, method: "ThresholdDetector.Method" = Method.FLOOR,
                 block_size=None, engine=None):
        if block_size is not None:
            warnings.warn("The `block_size` argument is deprecated and will be removed in v0.8.",
                          DeprecationWarning, stacklevel=2)
        super().__init__()
        self.threshold = int(threshold)
        self.method = ThresholdDetector.Method(method)
        self.fade_bias = fade_bias
        self.min_scene_len = min_scene_len
        self.processed_frame = False
        self.last_scene_cut: FrameTimecode | None = None
        self.add_final_scene = add_final_scene
        self.last_fade = {"frame": None, "type": None}
        self._metric_keys = [ThresholdDetector.THRESHOLD_VALUE_KEY]
        self._scorer = FrameScorer(engine)

    def get_metrics(self) -> list[str]:
        return self._metric_keys

    def score_flags(self) -> int:
        return _native.SCORE_BYTE_SUM

    def _faded_out(self, frame_avg: float) -> bool:
        if self.method == ThresholdDetector.Method.FLOOR:
            return frame_avg < self.threshold
        return frame_avg >= self.threshold

    def process_record(self, timecode: FrameTimecode, record, height: int, width: int) -> list[FrameTimecode]:
        if self.last_scene_cut is None:
            self.last_scene_cut = timecode
        cuts: list[FrameTimecode] = []
        stats = self.stats_manager
        if stats is not None and stats.metrics_exist(give_back(timecode), self._metric_keys):
            frame_avg = stats.get_metrics(give_back(timecode), self._metric_keys)[0]
        else:
            frame_avg = np.float64(int(record["byte_sum"]) / float(height * width * 3))
            if stats is not None:
                stats.set_metrics(give_back(timecode), {self._metric_keys[0]: frame_avg})
        if self.processed_frame:
            if self.last_fade["type"] == "in" and self._faded_out(frame_avg):
                self.last_fade["type"] = "out"
                self.last_fade["frame"] = timecode
            elif self.last_fade["type"] == "out" and not self._faded_out(frame_avg):
                if (timecode - self.last_scene_cut) >= self.min_scene_len:
                    f_out = self.last_fade["frame"]
                    duration = timecode.frame_num - f_out.frame_num
                    split = f_out.frame_num + round(duration * (1.0 + self.fade_bias) / 2.0)
                    cuts.append(new_like(timecode, split))
                    self.last_scene_cut = timecode
                self.last_fade["type"] = "in"
                self.last_fade["frame"] = timecode
        else:
            self.last_fade["frame"] = timecode
            # The very first frame is classified with `<` whatever the method (:161-165).
            self.last_fade["type"] = "out" if frame_avg < self.threshold else "in"
        self.processed_frame = True
        return cuts

    @plug_in_api
    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        stats = self.stats_manager
        if stats is not None and stats.metrics_exist(give_back(timecode), self._metric_keys):
            return self.process_record(timecode, None, frame_img.shape[0], frame_img.shape[1])
        record = self._scorer.score(frame_img, self.score_flags())
        return self.process_record(timecode, record, frame_img.shape[0], frame_img.shape[1])

    @plug_in_api
    def post_process(self, timecode: FrameTimecode) -> list[FrameTimecode]:
        cuts: list[FrameTimecode] = []
        elapsed = timecode if self.last_scene_cut is None else timecode - self.last_scene_cut
        if (self.last_fade["type"] == "out" and self.add_final_scene and self.last_fade["frame"] is not None
                and elapsed >= self.min_scene_len):
            cuts.append(self.last_fade["frame"])
        return cuts
