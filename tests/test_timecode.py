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


# ---- differential cases evaluated by the reference's own FrameTimecode (oracle/gen_timecode_golden.py) ----------

import json
import os
from fractions import Fraction

import pytest

from pyscenedetect_amd.timecode import FrameTimecode as _TC
from pyscenedetect_amd.timecode import framerate_to_fraction as _f2f

_CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "timecode_cases.json")))


def _fps(f):
    return Fraction(f[0], f[1]) if isinstance(f, list) else f


def _field(fn):
    try:
        return fn()
    except Exception as ex:  # noqa: BLE001
        return "raises:" + type(ex).__name__


def _describe(tc):
    return {"frame_num": _field(lambda: tc.frame_num), "seconds": _field(lambda: tc.seconds),
            "timecode": _field(tc.get_timecode), "timecode_p1": _field(lambda: tc.get_timecode(precision=1)),
            "timecode_p0": _field(lambda: tc.get_timecode(precision=0)),
            "timecode_trunc": _field(lambda: tc.get_timecode(use_rounding=False)), "hash": _field(lambda: hash(tc)),
            "int": _field(lambda: int(tc)), "float": _field(lambda: float(tc)), "str": _field(lambda: str(tc))}


def _outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001
        return {"raises": type(ex).__name__}


def _same(got, want, tag):
    if "raises" in want:
        assert got == want, f"{tag}: reference raises {want['raises']}, got {got}"
        return
    assert "ok" in got, f"{tag}: reference gives {want['ok']}, got {got}"
    if not isinstance(want["ok"], dict):
        assert got["ok"] == want["ok"], f"{tag}: {got['ok']!r} != {want['ok']!r}"
        return
    for key, w in want["ok"].items():
        g = got["ok"][key]
        if isinstance(w, float):
            assert g == pytest.approx(w, rel=0, abs=1e-9), f"{tag}: {key} {g!r} != {w!r}"
        else:
            assert g == w, f"{tag}: {key} {g!r} != {w!r}"


def test_construction_matches_reference():
    for c in _CASES["construct"]:
        v, f = c["value"], c["fps"]
        if isinstance(v, str) and v.startswith("copy:"):
            got = _outcome(lambda: _describe(_TC(_TC(int(v[5:]), _fps(f)))))
        elif isinstance(v, str) and v.startswith("fps_from_tc:"):
            got = _outcome(lambda: _describe(_TC(int(v[12:]), _TC(100, _fps(f)))))
        else:
            got = _outcome(lambda: _describe(_TC(v, _fps(f))))
        _same(got, {k: c[k] for k in ("ok", "raises") if k in c}, f"FrameTimecode({v!r}, {f!r})")


def test_invalid_construction_matches_reference():
    for c in _CASES["bad"]:
        v = complex(c["value"]) if c["is_complex"] else c["value"]
        got = _outcome(lambda: _describe(_TC(v, 25.0)))
        _same(got, {k: c[k] for k in ("ok", "raises") if k in c}, f"FrameTimecode({v!r})")


def test_arithmetic_matches_reference():
    for c in _CASES["arith"]:
        def run():
            x = _TC(c["a"], _fps(c["fps"]))
            y = _TC(c["b"], _fps(c["fps"])) if c["kind"] == "tc" else c["b"]
            r = x + y if c["op"] == "add" else x - y
            return {"frame_num": r.frame_num, "seconds": r.seconds}
        _same(_outcome(run), {k: c[k] for k in ("ok", "raises") if k in c},
              f"{c['a']!r} {c['op']} {c['b']!r} ({c['kind']}) @ {c['fps']!r}")


def test_comparisons_match_reference():
    for c in _CASES["compare"]:
        def run():
            x = _TC(c["a"], _fps(c["fps"]))
            y = _TC(c["b"], _fps(c["fps"])) if c["kind"] == "tc" else c["b"]
            return {"eq": x == y, "ne": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}
        _same(_outcome(run), {k: c[k] for k in ("ok", "raises") if k in c},
              f"{c['a']!r} vs {c['b']!r} ({c['kind']}) @ {c['fps']!r}")


def test_framerate_to_fraction_matches_reference():
    for c in _CASES["framerate"]:
        def run():
            q = _f2f(c["fps"])
            return [q.numerator, q.denominator]
        _same(_outcome(run), {k: c[k] for k in ("ok", "raises") if k in c}, f"framerate_to_fraction({c['fps']!r})")


def test_cross_rate_behaviour_matches_reference():
    for c in _CASES["cross_rate"]:
        def run():
            a, b = _TC(10, 25.0), _TC(10, 30.0)
            op = c["op"]
            return {"eq": a == b} if op == "eq" else {"frame_num": (a + b).frame_num} if op == "add" else {"lt": a < b}
        _same(_outcome(run), {k: c[k] for k in ("ok", "raises") if k in c}, f"cross-rate {c['op']}")


# ---- presentation-timestamp backed timecodes (same fixture file) ---------------------------------------------

from pyscenedetect_amd.timecode import Timecode as _PTS


def _mk(x, f):
    if isinstance(x, list) and x[0] == "pts":
        return _TC(_PTS(x[1], Fraction(*x[2])), _fps(f))
    if isinstance(x, list) and x[0] == "bare":
        return _PTS(x[1], Fraction(*x[2]))
    if isinstance(x, list) and x[0] == "tc":
        return _TC(x[1], _fps(f))
    return x


def test_pts_backed_views_match_reference():
    for c in _CASES["pts_describe"]:
        def d():
            tc = _mk(c["a"], c["fps"])
            r = _describe(tc)
            r.update({"pts": _field(lambda: tc.pts),
                      "time_base": _field(lambda: [tc.time_base.numerator, tc.time_base.denominator]),
                      "repr": _field(lambda: repr(tc)), "tc_exact": _field(lambda: tc.get_timecode(nearest_frame=False))})
            return r
        _same(_outcome(d), {k: c[k] for k in ("ok", "raises") if k in c}, f"{c['a']} @ {c['fps']}")


def test_pts_backed_arithmetic_matches_reference():
    for c in _CASES["pts_arith"]:
        def run():
            x, y = _mk(c["a"], c["fps"]), _mk(c["b"], c["fps"])
            r = x + y if c["op"] == "add" else x - y
            return {"frame_num": r.frame_num, "seconds": r.seconds, "pts": r.pts,
                    "time_base": [r.time_base.numerator, r.time_base.denominator], "repr": repr(r)}
        _same(_outcome(run), {k: c[k] for k in ("ok", "raises") if k in c}, f"{c['a']} {c['op']} {c['b']} @ {c['fps']}")


def test_pts_backed_comparisons_match_reference():
    for c in _CASES["pts_compare"]:
        def run():
            x, y = _mk(c["a"], c["fps"]), _mk(c["b"], c["fps"])
            return {"eq": x == y, "ne": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}
        _same(_outcome(run), {k: c[k] for k in ("ok", "raises") if k in c}, f"{c['a']} vs {c['b']} @ {c['fps']}")
    x = _TC(_PTS(1001, Fraction(1, 30000)), 25.0)
    y = _TC(_PTS(1001, Fraction(1, 30000)), 30.0)
    assert {"ok": {"eq": x == y, "lt": x < y, "hash_x": hash(x), "hash_y": hash(y)}} == _CASES["pts_cross_rate"]


def test_numpy_integers_are_frame_numbers_like_in_the_reference():
    """``FrameTimecode(frame, fps)`` with a frame number that came out of an array (``numpy.int64``): the reference takes whatever is
    not a string, float or timecode for a frame number (``common.py:259-264``); the mirror refused numpy integers with TypeError
    (found when the fuzzers started to compare exception TEXTS: ``FrameTimecode(None, fps)`` fails in the reference's ``timecode < 0``)."""
    import numpy as np

    from pyscenedetect_amd import FrameTimecode

    cuts = np.array([0, 15, 198, 377])
    tcs = [FrameTimecode(c, 25.0) for c in cuts]
    assert [int(t.frame_num) for t in tcs] == [0, 15, 198, 377]
    assert tcs[2].get_timecode() == "00:00:07.920" and tcs[1] < tcs[2] and (tcs[3] - tcs[2]).frame_num == 179
    assert FrameTimecode(np.int32(7), 30.0) == 7
    with pytest.raises(ValueError):
        FrameTimecode(np.int64(-1), 25.0)
    with pytest.raises(TypeError, match="not supported between"):
        FrameTimecode(None, 25.0)


def test_timecodes_survive_pickle_and_deepcopy():
    """Results cross process boundaries (one process per GPU, cut lists gathered on every rank): a timecode of each backing comes back
    equal, with the same hash and the same text."""
    import copy
    import pickle

    from pyscenedetect_amd import Timecode

    for tc in (FrameTimecode(5, 25.0), FrameTimecode(1.5, 29.97), FrameTimecode(Timecode(1500, Fraction(1, 1000)), 24.0)):
        for twin in (pickle.loads(pickle.dumps(tc)), copy.deepcopy(tc)):
            assert twin == tc and hash(twin) == hash(tc) and twin.get_timecode() == tc.get_timecode() and twin.frame_rate == tc.frame_rate
