"""Frame-accurate time values for the host side of the scoring engine.

Mirror of ``scenedetect.common.FrameTimecode`` / ``Timecode`` (reference ``scenedetect/common.py:163-174,191-860``) as
the detector hot path sees them.  A position is held in one of three forms, always next to a rational frame rate:

* ``"frames"``  an exact frame number (what constant-frame-rate decoders hand out);
* ``"seconds"`` a float number of seconds (``float`` / ``"HH:MM:SS.nnn"`` / ``"1.5s"`` inputs);
* ``"pts"``     a presentation timestamp ``pts * time_base`` (:class:`Timecode`; what PyAV-style decoders of variable
  frame rate video hand out).

Semantics follow the reference for every pair of forms:

* comparing with an ``int`` compares frame numbers; two timecodes compare by frame number, except two PTS-backed ones
  of the same nominal rate, which compare as exact rationals (``common.py:822-860``); a float/str operand is compared
  in seconds when this side is seconds- or PTS-backed, else converted with ``round(seconds * fps)``;
* ``a + b`` / ``a - b`` keep the form of the PTS-backed operand (finer time base when both are), else of ``a``; results
  clamp at zero (``common.py:640-760``);
* the hash is the frame number, so an ``int`` can index a stats dictionary (``common.py:783-791``).
"""

import warnings
from dataclasses import dataclass
from fractions import Fraction

MAX_FPS_DELTA = 1.0 / 1000000000.0
_NTSC_TOLERANCE = 1e-3

# The type names the reference's `scenedetect.common` exports (common.py:77-101), for code that annotates with them:
TimecodeLike = "int | float | str | Timecode | FrameTimecode"   # whatever FrameTimecode() accepts as a time
FrameRate = float | Fraction
CropRegion = tuple[int, int, int, int]                          # (x0, y0, x1, y1)
CutList = list["FrameTimecode"]
SceneList = list[tuple["FrameTimecode", "FrameTimecode"]]
TimecodePair = tuple["FrameTimecode", "FrameTimecode"]


@dataclass(frozen=True)
class Timecode:
    """Presentation timestamp of a frame: ``pts`` ticks of ``time_base`` seconds (reference ``common.py:163-174``)."""

    pts: int
    time_base: Fraction

    @property
    def seconds(self) -> float:
        return float(self.time_base * self.pts)


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
        return secs + ((hrs * 60 * 60) + (mins * 60))      # (the reference's order, common.py:523: the integer part summed first)
    if text.endswith("s"):
        text = text[:-1]
    if not text.replace(".", "").isdigit():
        raise ValueError("All characters in timecode seconds string must be digits.")
    return float(text)


class FrameTimecode:
    """A position in a video: frame number, seconds, or presentation timestamp, plus the frame rate."""

    __slots__ = ("_frames", "_secs", "_pts", "_rate")

    def __init__(self, timecode, fps=None):
        if _is_foreign_timecode(timecode):
            timecode = _adopt_foreign(timecode)
        elif not isinstance(timecode, (FrameTimecode, Timecode, int, float, str)) and hasattr(timecode, "pts") and hasattr(timecode, "time_base"):
            timecode = Timecode(int(timecode.pts), Fraction(timecode.time_base))      # another library's bare Timecode
        if isinstance(timecode, FrameTimecode):
            self._frames, self._secs, self._pts = timecode._frames, timecode._secs, timecode._pts
            self._rate = timecode._rate if fps is None else _as_rate(fps)
            return
        if fps is None:
            raise TypeError("fps is a required argument.")
        self._rate = _as_rate(fps)
        self._frames = None
        self._secs = None
        self._pts = None
        if isinstance(timecode, Timecode):
            self._pts = timecode
            return
        if isinstance(timecode, str) and timecode.isdigit():
            timecode = int(timecode)
        if isinstance(timecode, str):
            self._secs = parse_timecode_seconds(timecode, self._rate)
        elif isinstance(timecode, float):
            if timecode < 0.0:
                raise ValueError("Timecode frame number must be positive and greater than zero.")
            self._secs = timecode
        else:
            # Whatever is left is taken for a frame number, as in the reference (common.py:259-264: "only `int` remains"): a numpy
            # integer from an array of cuts is one, and what cannot be compared with 0 fails in that comparison (TypeError).
            if timecode < 0:
                raise ValueError("Timecode frame number must be positive and greater than zero.")
            self._frames = timecode

    # -- views ---------------------------------------------------------------------------------
    @property
    def frame_rate(self) -> Fraction:
        return self._rate

    @property
    def frame_num(self) -> int:
        """Frame number; for PTS-backed values an approximation from the nominal rate."""
        if self._pts is not None:
            return round(self._pts.seconds * float(self._rate))
        if self._frames is not None:
            return self._frames
        return round(self._secs * self._rate)

    @property
    def seconds(self) -> float:
        if self._pts is not None:
            return self._pts.seconds
        if self._secs is not None:
            return self._secs
        return float(self._frames / self._rate)

    @property
    def time_base(self) -> Fraction:
        return self._pts.time_base if self._pts is not None else 1 / self._rate

    @property
    def pts(self) -> int:
        return self._pts.pts if self._pts is not None else self.frame_num

    def equal_frame_rate(self, other) -> bool:
        if isinstance(other, FrameTimecode):
            other = other._rate
        if other is self._rate:          # (positions of one stream share their rate object: no float arithmetic for the usual case)
            return True
        return abs(float(self._rate) - float(other)) < MAX_FPS_DELTA

    # -- names the reference still answers to, each with one DeprecationWarning (common.py:292-306, 325-350, 375-383, 396-414)
    def _legacy(self, old: str, new: str) -> None:
        warnings.warn(f"{old} is deprecated; use {new} instead.", DeprecationWarning, stacklevel=3)

    @property
    def framerate(self) -> float:
        self._legacy("`framerate`", "`frame_rate`")
        return float(self._rate)

    def get_framerate(self) -> float:
        self._legacy("get_framerate()", "the `frame_rate` property")
        return float(self._rate)

    def get_frames(self) -> int:
        self._legacy("get_frames()", "the `frame_num` property")
        return self.frame_num

    def get_seconds(self) -> float:
        self._legacy("get_seconds()", "the `seconds` property")
        return self.seconds

    def equal_framerate(self, fps) -> bool:
        self._legacy("`equal_framerate()`", "`equal_frame_rate()`")
        return self.equal_frame_rate(fps)

    def get_timecode(self, precision: int = 3, use_rounding: bool = True, nearest_frame: bool = True) -> str:
        """``HH:MM:SS.nnn``; frame- and seconds-backed values snap to the frame boundary, a PTS already is one
        (reference ``common.py:421-465``)."""
        if nearest_frame and self._pts is None:
            secs = self.frame_num / float(self._rate)
        else:
            secs = self.seconds
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
        other = _own(other)
        if isinstance(other, int):
            return other
        if isinstance(other, float):
            return round(other * self._rate)
        if isinstance(other, str):
            return round(parse_timecode_seconds(other, self._rate) * self._rate)
        if isinstance(other, Timecode):
            return round(other.seconds * self._rate)
        if isinstance(other, FrameTimecode):
            if not self.equal_frame_rate(other._rate):
                raise ValueError("FrameTimecode instances require equal frame rate for frame-based arithmetic.")
            return other._frames if other._frames is not None else round(other.seconds * self._rate)
        raise TypeError("Cannot obtain frame number for this timecode.")

    def _seconds_of(self, other) -> float:
        other = _own(other)
        if isinstance(other, int):
            return float(other) / float(self._rate)
        if isinstance(other, float):
            return other
        if isinstance(other, str):
            return parse_timecode_seconds(other, self._rate)
        if isinstance(other, (Timecode, FrameTimecode)):
            return other.seconds
        raise TypeError("Unsupported type for performing arithmetic with FrameTimecode.")

    def _cmp_key(self, other):
        """(mine, theirs) in the unit the reference would compare in."""
        other = _own(other)
        if isinstance(other, FrameTimecode):
            if self._pts is not None and other._pts is not None and self._rate == other._rate:
                return self._pts.pts * self._pts.time_base, other._pts.pts * other._pts.time_base
            return self.frame_num, other.frame_num
        if isinstance(other, int):
            return self.frame_num, other
        if self._frames is None:
            return self.seconds, self._seconds_of(other)
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
        other = _own(other)
        out = FrameTimecode(self)
        theirs = other._pts if isinstance(other, FrameTimecode) else (other if isinstance(other, Timecode) else None)
        mine = self._pts
        if mine is not None and theirs is not None:
            base = mine.time_base
            a, b = mine.pts, theirs.pts
            if theirs.time_base != base:   # keep the finer of the two time bases
                base = min(mine.time_base, theirs.time_base)
                a = round(Fraction(mine.pts) * mine.time_base / base)
                b = round(Fraction(theirs.pts) * theirs.time_base / base)
            out._pts = Timecode(max(0, a + sign * b), base)
        elif mine is not None:
            ticks = round(self._seconds_of(other) / mine.time_base)
            out._pts = Timecode(max(0, mine.pts + sign * ticks), mine.time_base)
        elif theirs is not None:
            ticks = round(self.seconds / theirs.time_base)
            out._frames = out._secs = None
            out._pts = Timecode(max(0, theirs.pts + ticks if sign > 0 else ticks - theirs.pts), theirs.time_base)
        elif self._secs is not None:
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
        if self._pts is not None:
            return f"{self.get_timecode()} [pts={self._pts.pts}, time_base={self._pts.time_base}]"
        if self._secs is not None:
            return f"{self.get_timecode()} [seconds={self._secs}, fps={self._rate}]"
        return f"{self.get_timecode()} [frame_num={self._frames}, fps={self._rate}]"


def _is_foreign_timecode(obj) -> bool:
    """Another library's frame timecode (e.g. ``scenedetect.FrameTimecode``): duck-typed."""
    return not isinstance(obj, (FrameTimecode, Timecode)) and hasattr(obj, "frame_num") and hasattr(obj, "frame_rate")


def _own(obj):
    """An operand of another library's making as this package's: its FrameTimecode, or its bare ``Timecode(pts, time_base)``."""
    if isinstance(obj, (FrameTimecode, Timecode, int, float, str)) or obj is None:
        return obj
    if hasattr(obj, "frame_num") and hasattr(obj, "frame_rate"):
        return FrameTimecode(obj)
    if hasattr(obj, "pts") and hasattr(obj, "time_base"):
        return Timecode(int(obj.pts), Fraction(obj.time_base))
    return obj


def _adopt_foreign(obj) -> "FrameTimecode":
    """Take over a foreign timecode, keeping its presentation timestamp (or its seconds) when that is what it holds."""
    rate = framerate_to_fraction(obj.frame_rate)
    inner = getattr(obj, "_time", None)
    if inner is not None and hasattr(inner, "pts") and hasattr(inner, "time_base"):
        return FrameTimecode(Timecode(int(inner.pts), Fraction(inner.time_base)), rate)
    if inner is not None and type(inner).__name__ == "_Seconds" and isinstance(getattr(inner, "value", None), float):
        return FrameTimecode(float(inner.value), rate)          # (the reference's seconds-backed form, common.py:254-258)
    return FrameTimecode(int(obj.frame_num), rate)


class _AdoptedTimecode(FrameTimecode):
    """A foreign timecode taken over for a detector's own arithmetic; ``origin`` is the object it came from, which is what the
    caller gets back (a cut list must hold the caller's own kind of timecode: the reference's SceneManager sorts and compares
    the cuts of all its detectors with its stream's positions, scene_manager.py:403-408)."""

    __slots__ = ("origin",)


def adopt(timecode):
    """``timecode`` as one of this package's; a foreign one (``scenedetect.FrameTimecode``) remembers where it came from."""
    if isinstance(timecode, FrameTimecode) or not _is_foreign_timecode(timecode):
        return timecode
    own = _AdoptedTimecode(timecode)
    own.origin = timecode
    return own


def give_back(timecode):
    """The caller's own object for a timecode that was adopted; anything else as it is."""
    return getattr(timecode, "origin", timecode)


def new_like(timecode, frame_num: int):
    """Frame ``frame_num`` at the rate of ``timecode``, as the same kind of object (``FrameTimecode(n, fps=timecode)``)."""
    origin = getattr(timecode, "origin", None)
    if origin is not None:
        return adopt(type(origin)(frame_num, fps=origin))
    return FrameTimecode(frame_num, fps=timecode)


def _as_rate(fps) -> Fraction:
    if isinstance(fps, FrameTimecode):
        return fps._rate
    if _is_foreign_timecode(fps):
        return framerate_to_fraction(fps.frame_rate)
    if isinstance(fps, (float, Fraction)):
        return framerate_to_fraction(fps)
    raise TypeError(f"Wrong type for fps: {type(fps)} - expected float, Fraction, or FrameTimecode")
