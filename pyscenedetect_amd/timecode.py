"""Frame-accurate time values for the host side of the scoring engine.

Mirror of the subset of ``scenedetect.common.FrameTimecode`` (reference
``scenedetect/common.py:191-810``) that the detector hot path touches: a position is either an
exact frame number or a number of seconds, always paired with a rational frame rate.  Compare /
subtract semantics follow the reference for those two representations:

* comparing with an ``int`` compares frame numbers; with a ``float``/``str`` the other side is
  first converted to frames with ``round(seconds * fps)`` (``common.py:480-486,535-556``);
* ``a - b`` clamps at zero (``common.py:700-755``);
* the hash is the frame number, so an ``int`` can index a stats dictionary (``common.py:783-791``).

PTS-backed (variable frame rate) timecodes come from video decoders, which are outside this
package's scope, and are not modelled.
"""

from fractions import Fraction

MAX_FPS_DELTA = 1.0 / 1000000000.0
_NTSC_TOLERANCE = 1e-3

TimecodeLike = "int | float | str | FrameTimecode"


def framerate_to_fraction(fps) -> Fraction:
    """Exact rational frame rate (reference ``common.py:126-145``): whole numbers stay whole,
    ``N*1000/1001`` NTSC rates are recognised, anything else is limited to a 10000 denominator."""
    if fps <= MAX_FPS_DELTA:
        raise ValueError("Framerate must be positive and greater than zero.")
    if isinstance(fps, Fraction):
        return fps
    if fps == int(fps):
        return Fraction(int(fps), 1)
    base = round(fps * 1001 / 1000)
    if base > 0 and abs(base * 1000 / 1001 - fps) < _NTSC_TOLERANCE:
        return Fraction(base * 1000, 1001)
    return Fraction(fps).limit_denominator(10000)


def parse_timecode_seconds(text: str, rate: Fraction) -> float:
    """``'HH:MM:SS[.nnn]'``, ``'MM:SS[.nnn]'``, ``'123'`` (frames), ``'1.5'`` / ``'1.5s'`` -> seconds
    (reference ``common.py:488-533``)."""
    text = text.strip()
    if text.isdigit():
        return int(text) / float(rate)
    if ":" in text:
        parts = text.split(":")
        if len(parts) not in (2, 3):
            raise ValueError("Invalid timecode (too many separators).")
        hrs = int(parts[0]) if len(parts) == 3 else 0
        mins = int(parts[-2])
        secs = float(parts[-1]) if "." in parts[-1] else int(parts[-1])
        if not (hrs >= 0 and mins >= 0 and secs >= 0 and mins < 60 and secs < 60):
            raise ValueError("Invalid timecode range (values outside allowed range).")
        return secs + hrs * 3600 + mins * 60
    if text.endswith("s"):
        text = text[:-1]
    if not text.replace(".", "").isdigit():
        raise ValueError("All characters in timecode seconds string must be digits.")
    return float(text)


class FrameTimecode:
    """A position in a constant-frame-rate video: frame number or seconds, plus the frame rate."""

    __slots__ = ("_frames", "_secs", "_rate")

    def __init__(self, timecode, fps=None):
        if _is_foreign_timecode(timecode):
            timecode = FrameTimecode(int(timecode.frame_num), framerate_to_fraction(timecode.frame_rate))
        if isinstance(timecode, FrameTimecode):
            self._frames, self._secs = timecode._frames, timecode._secs
            self._rate = timecode._rate if fps is None else _as_rate(fps)
            return
        if fps is None:
            raise TypeError("fps is a required argument.")
        self._rate = _as_rate(fps)
        self._frames = None
        self._secs = None
        if isinstance(timecode, str) and timecode.isdigit():
            timecode = int(timecode)
        if isinstance(timecode, str):
            self._secs = parse_timecode_seconds(timecode, self._rate)
        elif isinstance(timecode, float):
            if timecode < 0.0:
                raise ValueError("Timecode frame number must be positive and greater than zero.")
            self._secs = timecode
        elif isinstance(timecode, int):
            if timecode < 0:
                raise ValueError("Timecode frame number must be positive and greater than zero.")
            self._frames = timecode
        else:
            raise TypeError(f"unsupported timecode type {type(timecode)}")

    # -- views ---------------------------------------------------------------------------------
    @property
    def frame_rate(self) -> Fraction:
        return self._rate

    @property
    def frame_num(self) -> int:
        if self._frames is not None:
            return self._frames
        return round(self._secs * self._rate)

    @property
    def seconds(self) -> float:
        if self._secs is not None:
            return self._secs
        return float(self._frames / self._rate)

    def get_timecode(self, precision: int = 3, use_rounding: bool = True) -> str:
        """``HH:MM:SS.nnn`` snapped to the frame boundary (reference ``common.py:421-465``)."""
        secs = self.frame_num / float(self._rate)
        hrs = int(secs / 3600.0)
        secs -= hrs * 3600.0
        mins = int(secs / 60.0)
        secs = max(0.0, secs - mins * 60.0)
        if use_rounding:
            secs = round(secs, precision)
        secs = min(60.0, secs)
        if int(secs) == 60:
            secs = 0.0
            mins += 1
            if mins >= 60:
                mins = 0
                hrs += 1
        msec = format(secs, f".{precision + 1}f") if precision else ""
        return f"{hrs:02d}:{mins:02d}:{int(secs):02d}{msec[-(2 + precision):-1]}"

    # -- conversions of the other operand --------------------------------------------------------
    def _frames_of(self, other) -> int:
        if _is_foreign_timecode(other):
            other = FrameTimecode(other)
        if isinstance(other, int):
            return other
        if isinstance(other, float):
            return round(other * self._rate)
        if isinstance(other, str):
            return round(parse_timecode_seconds(other, self._rate) * self._rate)
        if isinstance(other, FrameTimecode):
            if other._rate != self._rate and abs(float(other._rate) - float(self._rate)) > MAX_FPS_DELTA:
                raise ValueError("FrameTimecode instances require equal frame rate for frame-based arithmetic.")
            return other._frames if other._frames is not None else round(other._secs * self._rate)
        raise TypeError("Cannot obtain frame number for this timecode.")

    def _seconds_of(self, other) -> float:
        if _is_foreign_timecode(other):
            other = FrameTimecode(other)
        if isinstance(other, int):
            return float(other) / float(self._rate)
        if isinstance(other, float):
            return other
        if isinstance(other, str):
            return parse_timecode_seconds(other, self._rate)
        if isinstance(other, FrameTimecode):
            return other.seconds
        raise TypeError("Unsupported type for performing arithmetic with FrameTimecode.")

    def _cmp_key(self, other):
        """(mine, theirs) in the unit the reference would compare in."""
        if _is_foreign_timecode(other):
            other = FrameTimecode(other)
        if isinstance(other, FrameTimecode):
            return self.frame_num, other.frame_num
        if isinstance(other, int):
            return self.frame_num, other
        if self._secs is not None:
            return self._secs, self._seconds_of(other)
        return self._frames, self._frames_of(other)

    def __eq__(self, other):
        if other is None:
            return False
        a, b = self._cmp_key(other)
        return a == b

    def __ne__(self, other):
        return not self.__eq__(other)

    def __lt__(self, other):
        a, b = self._cmp_key(other)
        return a < b

    def __le__(self, other):
        a, b = self._cmp_key(other)
        return a <= b

    def __gt__(self, other):
        a, b = self._cmp_key(other)
        return a > b

    def __ge__(self, other):
        a, b = self._cmp_key(other)
        return a >= b

    def __hash__(self):
        return self.frame_num

    # -- arithmetic --------------------------------------------------------------------------------
    def _shifted(self, other, sign: int) -> "FrameTimecode":
        out = FrameTimecode(self)
        if self._secs is not None:
            if isinstance(other, FrameTimecode) and other._secs is not None:
                delta = other._secs
            else:
                delta = self._seconds_of(other)
            out._secs = max(0.0, self._secs + sign * delta)
        else:
            out._frames = max(0, self._frames + sign * self._frames_of(other))
        return out

    def __add__(self, other):
        return self._shifted(other, +1)

    def __sub__(self, other):
        return self._shifted(other, -1)

    def __int__(self):
        return self.frame_num

    def __float__(self):
        return self.seconds

    def __str__(self):
        return self.get_timecode()

    def __repr__(self):
        if self._secs is not None:
            return f"{self.get_timecode()} [seconds={self._secs}, fps={self._rate}]"
        return f"{self.get_timecode()} [frame_num={self._frames}, fps={self._rate}]"


def _is_foreign_timecode(obj) -> bool:
    """Another library's frame timecode (e.g. ``scenedetect.FrameTimecode``): duck-typed."""
    return not isinstance(obj, FrameTimecode) and hasattr(obj, "frame_num") and hasattr(obj, "frame_rate")


def _as_rate(fps) -> Fraction:
    if isinstance(fps, FrameTimecode):
        return fps._rate
    if _is_foreign_timecode(fps):
        return framerate_to_fraction(fps.frame_rate)
    if isinstance(fps, (float, Fraction)):
        return framerate_to_fraction(fps)
    raise TypeError(f"Wrong type for fps: {type(fps)} - expected float, Fraction, or FrameTimecode")


def min_len_frames(length, rate: Fraction) -> int:
    """Frames that ``(a - b) >= length`` compares against for frame-number timecodes."""
    if isinstance(length, int):
        return length
    if isinstance(length, float):
        return round(length * rate)
    if isinstance(length, str):
        if length.strip().isdigit():
            return int(length.strip())
        return round(parse_timecode_seconds(length, rate) * rate)
    if isinstance(length, FrameTimecode):
        return length.frame_num
    raise TypeError(f"unsupported min_scene_len type {type(length)}")
