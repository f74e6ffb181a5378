"""Frame sources for the scoring engine.

Decoding video is outside this package's scope (the reference's
``VideoStream`` backends wrap cv2/PyAV/MoviePy decoders).  What the hot path needs from a source
is the small contract ``SceneManager.detect_scenes`` uses (reference
``scenedetect/video_stream.py:79-222``): ``frame_size``, ``frame_rate``, ``base_timecode``,
``position``, ``frame_number``, ``duration`` and ``read()``.  Any object with those members works,
including the reference's own streams; :class:`ArrayVideoStream` serves decoded BGR frames from
memory (tests, benchmarks, frames handed over by an external decoder).
"""

from fractions import Fraction

import numpy as np

from pyscenedetect_amd.timecode import FrameTimecode, Timecode


class ArrayVideoStream:
    """In-memory BGR frames ``uint8[N,H,W,3]`` (or any indexable of ``uint8[H,W,3]``)."""

    def __init__(self, frames, fps: float | Fraction = 25.0, name: str = "array", pts=None, time_base: Fraction | None = None):
        """``pts`` / ``time_base``: optional presentation timestamps (one per frame), for frames that came out of a
        variable-frame-rate decoder; positions are then PTS-backed like those of the reference's PyAV backend."""
        if (pts is None) != (time_base is None):
            raise ValueError("pts and time_base go together")
        if pts is not None and len(pts) != len(frames):
            raise ValueError("one presentation timestamp per frame is required")
        self._pts = pts
        self._time_base = Fraction(time_base) if time_base is not None else None
        self._frames = frames
        self._n = len(frames)
        first = np.asarray(frames[0]) if self._n else np.zeros((0, 0, 3), np.uint8)
        self._size = (int(first.shape[1]), int(first.shape[0]))
        self._base = FrameTimecode(0, fps if isinstance(fps, (float, Fraction)) else float(fps))
        self._next = 0
        self.name = name

    @property
    def frame_size(self) -> tuple[int, int]:
        """(width, height)"""
        return self._size

    @property
    def frame_rate(self) -> Fraction:
        return self._base.frame_rate

    @property
    def base_timecode(self) -> FrameTimecode:
        return self._base

    @property
    def duration(self) -> FrameTimecode:
        if self._pts is not None and self._n:
            return FrameTimecode(Timecode(int(self._pts[-1]), self._time_base), self._base.frame_rate)
        return self._base + self._n

    @property
    def frame_number(self) -> int:
        """Number of frames read so far (0 before the first read)."""
        return self._next

    @property
    def position(self) -> FrameTimecode:
        """Timecode of the last frame read (frame 0 before any read)."""
        if self._pts is not None and self._n:
            return FrameTimecode(Timecode(int(self._pts[max(0, self._next - 1)]), self._time_base), self._base.frame_rate)
        return self._base + max(0, self._next - 1)

    def read(self, decode: bool = True):
        if self._next >= self._n:
            return False
        frame = self._frames[self._next]
        self._next += 1
        return np.asarray(frame) if decode else True

    def reset(self) -> None:
        self._next = 0

    def seek(self, target) -> None:
        frame = target.frame_num if isinstance(target, FrameTimecode) else int(target)
        if frame < 0:
            raise ValueError("Target seek position cannot be negative!")
        self._next = min(frame, self._n)
