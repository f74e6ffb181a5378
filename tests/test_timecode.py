"""CPU: FrameTimecode semantics the detectors rely on (reference scenedetect/common.py:191-810)."""
from fractions import Fraction

import pytest

from pyscenedetect_amd import FrameTimecode
from pyscenedetect_amd.timecode import framerate_to_fraction


def test_frame_and_seconds_views():
    tc = FrameTimecode(50, 25.0)
    assert tc.frame_num == 50 and tc.seconds == 2.0 and tc.get_timecode() == "00:00:02.000"
    assert FrameTimecode("00:01:00.500", 10.0).frame_num == 605
    assert FrameTimecode(1.5, 30.0).frame_num == 45
    assert FrameTimecode("90", 30.0).frame_num == 90
    assert framerate_to_fraction(23.976) == Fraction(24000, 1001)
    assert framerate_to_fraction(29.97) == Fraction(30000, 1001)
    with pytest.raises(ValueError):
        FrameTimecode(-1, 25.0)
    with pytest.raises(TypeError):
        FrameTimecode(1)


def test_compare_and_arithmetic_like_reference():
    a, b = FrameTimecode(100, 25.0), FrameTimecode(40, 25.0)
    assert (a - b).frame_num == 60 and (b - a).frame_num == 0        # clamps at zero
    assert (a - b) >= 60 and not (a - b) >= 61
    assert (a - b) >= 2.4 and (a - b) >= 2.41 and not (a - b) >= 2.43                      # seconds -> round(secs*fps) frames
    assert (a - b) >= "00:00:02.400" and (a - b) >= "60"
    assert a + 5 == 105 and a == FrameTimecode(4.0, 25.0)
    assert hash(a) == 100 and {a: 1}[100] == 1                       # int keys index a stats dict
    assert sorted([a, b])[0] is b
    with pytest.raises(ValueError):
        a - FrameTimecode(1, 30.0)


def test_foreign_timecode_is_accepted():
    class RefTimecode:  # shaped like scenedetect.FrameTimecode
        def __init__(self, n, rate):
            self.frame_num, self.frame_rate = n, rate

    mine = FrameTimecode(RefTimecode(30, Fraction(25, 1)))
    assert mine.frame_num == 30 and mine.frame_rate == 25
    assert FrameTimecode(7, fps=RefTimecode(0, Fraction(30000, 1001))).frame_rate == Fraction(30000, 1001)
    assert mine >= RefTimecode(30, Fraction(25, 1)) and (mine - RefTimecode(10, Fraction(25, 1))).frame_num == 20
