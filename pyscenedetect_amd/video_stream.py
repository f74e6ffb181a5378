"""Frame sources for the scoring engine.

Decoding video is outside this package's scope (the reference's ``VideoStream``
backends wrap cv2 / PyAV / MoviePy decoders).  What the hot path needs from a source is the small contract
``SceneManager.detect_scenes`` uses (reference ``scenedetect/video_stream.py:79-222``): ``frame_size``,
``frame_rate``, ``base_timecode``, ``position``, ``frame_number``, ``duration`` and ``read()``.  Any object with
those members works, including the reference's own streams.  :class:`VideoStream` states the whole interface of
the reference under the reference's names -- a decoder written against ``scenedetect.video_stream.VideoStream``
subclasses this one unchanged -- together with its three exceptions; :class:`ArrayVideoStream` implements it for
decoded BGR frames in memory (tests, benchmarks, frames handed over by an external decoder).
"""

import typing as ty
from abc import ABC, abstractmethod
from fractions import Fraction

import numpy as np

from pyscenedetect_amd.timecode import FrameTimecode, Timecode


class SeekError(Exception):
    """Seeking failed or the stream cannot seek (reference ``video_stream.py:43-49``); the stream stays usable, its
    position may have been reset."""


class VideoOpenFailure(Exception):
    """A backend could not open its input (reference ``video_stream.py:52-60``)."""

    def __init__(self, message: str = "Unknown backend error."):
        super().__init__(message)


class FrameRateUnavailable(VideoOpenFailure):
    """The frame rate of the input is unknown; the one message every backend uses (reference ``video_stream.py:63-71``)."""

    def __init__(self):
        super().__init__("Unable to obtain video framerate! Specify `frame_rate` manually, or"
                         " re-encode/re-mux the video and try again.")


class VideoStream(ABC):
    """What every frame source provides (reference ``scenedetect/video_stream.py:79-222``, same member names and meaning).

    ``position`` is the presentation time of the LAST frame read (0 before the first read and after it alike),
    ``frame_number`` the count of frames read so far; ``read()`` returns the next frame as ``uint8[H, W, 3]`` in B, G, R
    order (``detector.py:56``), ``False`` at the end; ``read(decode=False)`` only advances.  ``seek(n)`` makes frame
    n + 1 (1-based) the next one read."""

    BACKEND_NAME: ty.ClassVar[str]
    _decode_failures: int = 0

    @property
    def base_timecode(self) -> FrameTimecode:
        """A zero timecode at the stream's frame rate: the time base of everything derived from this stream."""
        return FrameTimecode(timecode=0, fps=self.frame_rate)

    @property
    def decode_failures(self) -> int:
        """Frames the backend could not decode and skipped (0 for backends that do not count them)."""
        return self._decode_failures

    @property
    @abstractmethod
    def path(self) -> str: ...

    @property
    @abstractmethod
    def name(self) -> str: ...

    @property
    @abstractmethod
    def is_seekable(self) -> bool: ...

    @property
    @abstractmethod
    def frame_rate(self) -> Fraction: ...

    @property
    @abstractmethod
    def duration(self) -> FrameTimecode | None: ...

    @property
    @abstractmethod
    def frame_size(self) -> tuple[int, int]: ...

    @property
    @abstractmethod
    def aspect_ratio(self) -> float: ...

    @property
    @abstractmethod
    def position(self) -> FrameTimecode: ...

    @property
    @abstractmethod
    def position_ms(self) -> float: ...

    @property
    @abstractmethod
    def frame_number(self) -> int: ...

    @abstractmethod
    def read(self, decode: bool = True) -> np.ndarray | bool: ...

    @abstractmethod
    def reset(self) -> None: ...

    @abstractmethod
    def seek(self, target) -> None: ...


class ArrayVideoStream(VideoStream):
    """In-memory BGR frames ``uint8[N,H,W,3]`` (or any indexable of ``uint8[H,W,3]``)."""

    BACKEND_NAME = "array"

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
        shape = getattr(frames, "shape", None)
        if shape is not None and len(shape) == 4:      # one array (or something shaped like one): no frame is read to learn the size
            self._size = (int(shape[2]), int(shape[1]))
        else:
            first = np.asarray(frames[0]) if self._n else np.zeros((0, 0, 3), np.uint8)
            self._size = (int(first.shape[1]), int(first.shape[0]))
        self._base = FrameTimecode(0, fps if isinstance(fps, (float, Fraction)) else float(fps))
        self._next = 0
        self._name = name

    @property
    def path(self) -> str:
        """There is no file behind the frames: the name stands in."""
        return self._name

    @property
    def name(self) -> str:
        return self._name

    @name.setter
    def name(self, value: str) -> None:
        self._name = value

    @property
    def is_seekable(self) -> bool:
        return True

    @property
    def aspect_ratio(self) -> float:
        return 1.0

    @property
    def position_ms(self) -> float:
        """Presentation time of the last frame read, in milliseconds (0.0 before and after the first read)."""
        return self.position.seconds * 1000.0

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
