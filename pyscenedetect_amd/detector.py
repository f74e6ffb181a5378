"""Detector plug-in interface and the minimum-scene-length filter.

Same surface as the reference's ``scenedetect/detector.py`` (``SceneDetector`` :37-103,
``FlashFilter`` :106-224) so a detector written against PySceneDetect's plug-in API drops in.
What is new is :meth:`SceneDetector.process_record`: GPU-backed detectors split every
``process_frame`` into *pixel work* (done once per frame on the device, possibly for a whole
batch and for several detectors at once) and the *decision* made from the resulting integer
record.  ``process_frame(timecode, frame)`` remains and is just "score one frame, then decide".
"""

import math
from abc import ABC, abstractmethod
from enum import Enum

import numpy as np

from pyscenedetect_amd.timecode import FrameTimecode, Timecode, adopt, give_back, parse_timecode_seconds


def plug_in_api(method):
    """``process_frame`` / ``post_process`` of the built-in detectors as ANY caller of the plug-in API may use them (reference
    ``detector.py:48-73``) -- the reference's own SceneManager included, whose timecodes are ``scenedetect.FrameTimecode`` objects:
    those are adopted for the detector's arithmetic and every cut goes back as the caller's own object."""
    import functools

    @functools.wraps(method)
    def call(self, timecode, *args, **kwargs):
        own = adopt(timecode)
        cuts = method(self, own, *args, **kwargs)
        return cuts if own is timecode else [give_back(cut) for cut in cuts]

    return call


class SceneDetector(ABC):
    """Base class of all detectors (reference ``detector.py:37-103``)."""

    def __init__(self):
        self._stats_manager = None

    @abstractmethod
    def process_frame(self, timecode: FrameTimecode, frame_img: np.ndarray) -> list[FrameTimecode]:
        """Process the next frame (24-bit BGR image); returns the cuts detected, if any."""

    def post_process(self, timecode: FrameTimecode) -> list[FrameTimecode]:
        """Called once after the last frame."""
        return []

    @property
    def event_buffer_length(self) -> int:
        """How many frames in the past an emitted cut can lie."""
        return 0

    @property
    def stats_manager(self):
        return self._stats_manager

    @stats_manager.setter
    def stats_manager(self, value):
        self._stats_manager = value

    def get_metrics(self) -> list[str]:
        return []

    # -- extension used by the batched GPU path --------------------------------------------------
    def score_flags(self) -> int:
        """Bit-or of ``PSD_SCORE_*`` terms this detector needs per frame (0 = not GPU-backed)."""
        return 0

    def edge_kernel_size(self) -> int:
        """Dilation size for the edge term, 0 = derive from the frame size."""
        return 0

    def process_record(self, timecode: FrameTimecode, record, height: int, width: int) -> list[FrameTimecode]:
        """Decide from a precomputed per-frame score record (``psd_frame_scores``)."""
        raise NotImplementedError

    def hash_thumb_size(self) -> int:
        """Side of the grey INTER_AREA thumbnail this detector needs per frame (0 = none)."""
        return 0

    def process_thumb(self, timecode: FrameTimecode, thumb) -> list[FrameTimecode]:
        """Decide from a precomputed thumbnail (``psd_hash_thumbs*``)."""
        raise NotImplementedError


def _time_backed(timecode) -> bool:
    """A position held as a presentation timestamp or as seconds (not as a frame number)."""
    return timecode is not None and getattr(timecode, "_frames", 0) is None      # (this package's timecodes: exactly one backing is set)


class FlashFilter:
    """Enforces a minimum scene length on a stream of above/below-threshold decisions.

    MERGE collapses bursts of cuts closer than ``length`` into the last one of the burst (emitted
    late, once the burst is over); SUPPRESS drops cuts until ``length`` has passed since the last
    one.  Behaviour follows the reference (``detector.py:160-224``), including its quirks: the
    filter arms only after the first emitted cut, and an integer length is converted to seconds
    with the first frame's rate and back to frames for every comparison.
    """

    class Mode(Enum):
        MERGE = 0
        SUPPRESS = 1

    def __init__(self, mode: "FlashFilter.Mode", length):
        self._mode = mode
        self._length_frames = 0
        self._length_secs = None
        if isinstance(length, float):
            self._length_secs = length
        elif isinstance(length, str) and not length.strip().isdigit():
            self._length_secs = parse_timecode_seconds(length, 100)
        elif isinstance(length, (Timecode, FrameTimecode)) or (hasattr(length, "seconds") and not isinstance(length, (int, str))):
            self._length_secs = length.seconds       # any TimecodeLike (reference detector.py:136-137), of whichever library
        else:
            self._length_frames = int(length)
        self._threshold_frames = None  # resolved on the first frame
        self._last_above = None
        self._armed = False
        self._merging = False
        self._merge_start = None

    @property
    def max_behind(self) -> int:
        if self._mode == FlashFilter.Mode.SUPPRESS:
            return 0
        if self._length_secs is not None:
            return math.ceil(self._length_secs * 240.0)
        return self._length_frames

    @property
    def _disabled(self) -> bool:
        if self._length_secs is not None:
            return self._length_secs <= 0.0
        return self._length_frames <= 0

    def _resolve(self, timecode: FrameTimecode) -> int:
        if self._threshold_frames is None:
            rate = timecode.frame_rate
            secs = self._length_secs
            if secs is None:
                secs = self._length_frames / float(rate)
                self._length_secs = secs
            self._threshold_frames = round(secs * rate)
        return self._threshold_frames

    def _filter_by_time(self, timecode: FrameTimecode, above_threshold: bool) -> list[FrameTimecode]:
        """Positions that are presentation timestamps (variable frame rate): the reference's own comparisons,
        ``(timecode - last_above) >= seconds`` on timecode arithmetic (``detector.py:171-224``) -- frame numbers derived
        from an average rate would misplace the gaps."""
        if self._length_secs is None:
            self._length_secs = self._length_frames / float(timecode.frame_rate)
        if self._last_above is None:
            self._last_above = timecode
        min_length_met = (timecode - self._last_above) >= self._length_secs
        if self._mode == FlashFilter.Mode.SUPPRESS:
            if not (above_threshold and min_length_met):
                return []
            self._last_above = timecode
            return [timecode]
        if self._mode != FlashFilter.Mode.MERGE:
            raise RuntimeError("Unhandled FlashFilter mode.")
        if above_threshold:
            self._last_above = timecode
        if self._merging:
            if min_length_met and not above_threshold and (self._last_above - self._merge_start) >= self._length_secs:
                self._merging = False
                return [self._last_above]
            return []
        if not above_threshold:
            return []
        if min_length_met:
            self._armed = True
            return [timecode]
        if self._armed:
            self._merging = True
            self._merge_start = timecode
        return []

    def filter(self, timecode: FrameTimecode, above_threshold: bool) -> list[FrameTimecode]:
        if self._disabled:
            return [timecode] if above_threshold else []
        # positions that are not plain frame numbers -- presentation timestamps, or timecodes a caller of process_frame() built
        # from seconds -- are compared the way the reference compares everything, on timecode arithmetic (the frame-number path
        # below is its equivalent for frame-backed positions; tools/fuzz_host_vs_reference.py found the seconds-backed case)
        # (... and so is a frame-backed position that meets a time-backed one this filter kept from an earlier video: one manager
        #  on a VFR and then a CFR stream without clear(), seed 115 case 3309)
        if _time_backed(timecode) or _time_backed(self._last_above) or _time_backed(self._merge_start):
            return self._filter_by_time(timecode, above_threshold)
        need = self._resolve(timecode)
        now = timecode.frame_num
        if self._last_above is None:
            self._last_above = timecode
        gap_ok = max(0, now - self._last_above.frame_num) >= need
        if self._mode == FlashFilter.Mode.SUPPRESS:
            if above_threshold and gap_ok:
                self._last_above = timecode
                return [timecode]
            return []
        if self._mode != FlashFilter.Mode.MERGE:
            raise RuntimeError("Unhandled FlashFilter mode.")
        if above_threshold:
            self._last_above = timecode
        if self._merging:
            burst_len = max(0, self._last_above.frame_num - self._merge_start.frame_num)
            if gap_ok and not above_threshold and burst_len >= need:
                self._merging = False
                return [self._last_above]
            return []
        if not above_threshold:
            return []
        if gap_ok:
            self._armed = True
            return [timecode]
        if self._armed:
            self._merging = True
            self._merge_start = timecode
        return []
