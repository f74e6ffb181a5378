"""HashDetector on the MI355X scoring engine
(reference ``scenedetect/detectors/hash_detector.py:31-151``).

The device does the per-pixel half of ``hash_frame`` -- ``cv2.cvtColor(BGR2GRAY)`` and
``cv2.resize(INTER_AREA)`` to a ``size*lowpass`` square, bit-exact integer / float32 arithmetic
(``psd_hash_thumbs*``).  The 1 KiB per frame that is left (scale by the maximum, DCT, median threshold,
Hamming distance) runs on the host in ``psd_epilogue_hash_bits``.
"""

import numpy as np

from pyscenedetect_amd import epilogue
from pyscenedetect_amd.detector import SceneDetector, plug_in_api
from pyscenedetect_amd.timecode import FrameTimecode, give_back


def _thumbs_of(engine, frame_img: np.ndarray, size: int) -> np.ndarray:
    frame = np.asarray(frame_img)
    if frame.dtype != np.uint8:
        raise ValueError("Image must be 8-bit BGR")
    if frame.ndim != 3 or frame.shape[2] != 3:
        raise ValueError("Image must have three color channels")
    if engine is None:
        from pyscenedetect_amd.engine import default_engine

        engine = default_engine()
    return engine.hash_thumbs_host(frame[None], size)[0]


class HashDetector(SceneDetector):
    """Cuts where the perceptual hashes of adjacent frames differ in at least ``threshold`` of their bits."""

    def __init__(self, threshold: float = 0.35, size: int = 8, lowpass: int = 2, min_scene_len=15, engine=None):
        super().__init__()
        self._threshold = threshold
        self._min_scene_len = min_scene_len
        self._size = size
        self._size_sq = float(size * size)
        self._factor = lowpass
        self._last_scene_cut = None
        self._last_hash = None
        self._metric_key = f"hash_dist [size={self._size} lowpass={self._factor}]"
        self._engine = engine

    def get_metrics(self) -> list[str]:
        return [self._metric_key]

    def hash_thumb_size(self) -> int:
        return self._size * self._factor

    @staticmethod
    def hash_frame(frame_img: np.ndarray, hash_size: int, factor: int, engine=None) -> np.ndarray:
        """Perceptual hash bool[hash_size, hash_size] of one BGR frame (reference ``hash_detector.py:117-151``)."""
        thumb = _thumbs_of(engine, frame_img, hash_size * factor)
        return epilogue.hash_bits(thumb[None], hash_size)[0]

    @property
    def hash_size(self) -> int:
        return self._size

    def process_thumb(self, timecode: FrameTimecode, thumb: np.ndarray, bits: np.ndarray | None = None) -> list[FrameTimecode]:
        """Decide from the frame's grey ``size*lowpass`` square thumbnail (or from its hash ``bits`` when the caller
        already ran ``epilogue.hash_bits`` over a whole batch of thumbnails)."""
        cut_list = []
        if self._last_scene_cut is None:
            self._last_scene_cut = timecode
        curr_hash = bits if bits is not None else epilogue.hash_bits(np.asarray(thumb)[None], self._size)[0]
        if self._last_hash is not None:
            hash_dist = int(np.count_nonzero(curr_hash.flatten() != self._last_hash.flatten()))
            hash_dist_norm = hash_dist / self._size_sq
            if self.stats_manager is not None:
                self.stats_manager.set_metrics(give_back(timecode), {self._metric_key: hash_dist_norm})
            if hash_dist_norm >= self._threshold and ((timecode - self._last_scene_cut) >= self._min_scene_len):
                cut_list.append(timecode)
                self._last_scene_cut = timecode
        self._last_hash = curr_hash
        return cut_list

    @plug_in_api
    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        return self.process_thumb(timecode, _thumbs_of(self._engine, frame_img, self.hash_thumb_size()))
