"""Differential fixtures for FrameTimecode: every case is evaluated by the REFERENCE's own
``scenedetect.common.FrameTimecode`` / ``framerate_to_fraction`` (unmodified, imported from /root/reference over the cv2
shim) and the outcome -- value or exception type -- is stored.  tests/test_timecode.py replays the cases against
``pyscenedetect_amd.timecode``.

    PYTHONPATH=oracle/cv2_shim:/root/repo:/root/reference python oracle/gen_timecode_golden.py

TEST INFRASTRUCTURE ONLY.  The fixture travels to the GPU box; /root/reference does not.
"""
import json
import os
import sys
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "cv2_shim"), os.path.dirname(HERE), "/root/reference"]

from scenedetect.common import FrameTimecode, Timecode, framerate_to_fraction  # noqa: E402

FPS = [1.0, 10.0, 23.976, 24.0, 25.0, 29.97, 30.0, 59.94, 60.0, 120.0, [24000, 1001], [30000, 1001], [60000, 1001]]
VALUES = [0, 1, 10, 99, 1001, 86400, 0.0, 0.5, 1.0, 2.41, 10.0, 59.999, 3600.5, "0", "100", "00:00:01", "00:01:00.500",
          "01:00:00", "1:02:03.456", "12.5", "1.5s", "10s", "00:59.9", "00:00:00.001", "0.04", "23:59:59.999"]
BAD_VALUES = [-1, -0.5, "abc", "1:2:3:4", "00:61:00", "00:00:61", "-5", "5x", "", "1.5.5", None, [1], 1 + 2j]
OPERANDS = [0, 1, 15, 100, 0.0, 0.6, 1.0, 2.5, "15", "0.6s", "00:00:01.000", "00:00:00.500"]


def fps_of(f):
    return Fraction(f[0], f[1]) if isinstance(f, list) else f


def outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001 -- the exception TYPE is the fixture
        return {"raises": type(ex).__name__}


def _field(fn):
    try:
        return fn()
    except Exception as ex:  # noqa: BLE001
        return "raises:" + type(ex).__name__


def describe(tc):
    return {"frame_num": _field(lambda: tc.frame_num), "seconds": _field(lambda: tc.seconds),
            "timecode": _field(tc.get_timecode), "timecode_p1": _field(lambda: tc.get_timecode(precision=1)),
            "timecode_p0": _field(lambda: tc.get_timecode(precision=0)),
            "timecode_trunc": _field(lambda: tc.get_timecode(use_rounding=False)), "hash": _field(lambda: hash(tc)),
            "int": _field(lambda: int(tc)), "float": _field(lambda: float(tc)), "str": _field(lambda: str(tc))}


def main():
    out = {"construct": [], "bad": [], "arith": [], "compare": [], "framerate": []}
    for f in FPS:
        for v in VALUES:
            out["construct"].append({"value": v, "fps": f, **outcome(lambda: describe(FrameTimecode(v, fps_of(f))))})
    for v in BAD_VALUES:
        enc = repr(v) if isinstance(v, complex) else v
        out["bad"].append({"value": enc, "is_complex": isinstance(v, complex), "fps": 25.0,
                           **outcome(lambda: describe(FrameTimecode(v, 25.0)))})
    for f in (25.0, 29.97, [24000, 1001], 10.0):
        for a in (0, 10, 100, 1.0, 2.41, "00:00:10.000"):
            for b in OPERANDS:
                for kind in ("raw", "tc"):
                    def rhs():
                        return FrameTimecode(b, fps_of(f)) if kind == "tc" else b
                    for op in ("add", "sub"):
                        def run():
                            x = FrameTimecode(a, fps_of(f))
                            y = rhs()
                            r = x + y if op == "add" else x - y
                            return {"frame_num": r.frame_num, "seconds": r.seconds}
                        out["arith"].append({"a": a, "b": b, "fps": f, "kind": kind, "op": op, **outcome(run)})
                    def cmp():
                        x = FrameTimecode(a, fps_of(f))
                        y = rhs()
                        return {"eq": x == y, "ne": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}
                    out["compare"].append({"a": a, "b": b, "fps": f, "kind": kind, **outcome(cmp)})
    # copy constructor and fps taken from another timecode
    for f in (25.0, 29.97):
        base = FrameTimecode(100, fps_of(f))
        out["construct"].append({"value": "copy:100", "fps": f, **outcome(lambda: describe(FrameTimecode(base)))})
        out["construct"].append({"value": "fps_from_tc:50", "fps": f, **outcome(lambda: describe(FrameTimecode(50, base)))})
    for x in [1.0, 10.0, 23.976, 23.98, 24.0, 25.0, 29.97, 29.970029, 30.0, 59.94, 59.9, 60.0, 119.88, 120.0, 12.5, 14.985,
              7.4925, 15.0, 0.5, 1e-10, 0.0, -1.0, 1000.0, 33.333, 47.952]:
        def fr():
            q = framerate_to_fraction(x)
            return [q.numerator, q.denominator]
        out["framerate"].append({"fps": x, **outcome(fr)})
    # different frame rates in one expression
    def cross(op):
        a, b = FrameTimecode(10, 25.0), FrameTimecode(10, 30.0)
        return {"eq": a == b} if op == "eq" else {"frame_num": (a + b).frame_num} if op == "add" else {"lt": a < b}
    out["cross_rate"] = [{"op": op, **outcome(lambda: cross(op))} for op in ("eq", "add", "lt")]
    # ---- presentation-timestamp backed values (variable frame rate decoders) --------------------------------------
    # operand encoding: ["pts", pts, [num, den]] = FrameTimecode(Timecode(pts, Fraction(num, den)), fps);
    #                   ["bare", pts, [num, den]] = a bare Timecode; anything else is passed as is / wrapped as before
    def mk(x, f):
        if isinstance(x, list) and x[0] == "pts":
            return FrameTimecode(Timecode(x[1], Fraction(*x[2])), fps_of(f))
        if isinstance(x, list) and x[0] == "bare":
            return Timecode(x[1], Fraction(*x[2]))
        if isinstance(x, list) and x[0] == "tc":
            return FrameTimecode(x[1], fps_of(f))
        return x

    PTS = [["pts", 0, [1, 1000]], ["pts", 1, [1, 1000]], ["pts", 40, [1, 1000]], ["pts", 1001, [1, 30000]],
           ["pts", 2002, [1, 30000]], ["pts", 90000, [1, 90000]], ["pts", 93753, [1, 90000]], ["pts", 417, [1, 10000]],
           ["pts", 12345678, [1, 1000000]], ["pts", 3, [1001, 30000]]]
    OTHERS = PTS[:6] + [["bare", 40, [1, 1000]], ["bare", 3003, [1, 30000]], ["tc", 1], ["tc", 10], ["tc", 0.5],
                        ["tc", "00:00:01.500"], 0, 1, 25, 0.04, 1.0, "00:00:00.040", "2", "0.5s"]
    out["pts_describe"] = []
    for f in (25.0, 29.97, [30000, 1001], 1000.0):
        for a in PTS:
            def d():
                tc = mk(a, f)
                r = describe(tc)
                r.update({"pts": _field(lambda: tc.pts), "time_base": _field(lambda: [tc.time_base.numerator, tc.time_base.denominator]),
                          "repr": _field(lambda: repr(tc)), "tc_exact": _field(lambda: tc.get_timecode(nearest_frame=False))})
                return r
            out["pts_describe"].append({"a": a, "fps": f, **outcome(d)})
    out["pts_arith"], out["pts_compare"] = [], []
    for f in (25.0, [30000, 1001]):
        for a in PTS + [["tc", 5], ["tc", 0.2]]:
            for b in OTHERS:
                if not (isinstance(a, list) and a[0] == "pts") and not (isinstance(b, list) and b[0] in ("pts", "bare")):
                    continue   # no PTS involved: covered above
                for op in ("add", "sub"):
                    def run():
                        x, y = mk(a, f), mk(b, f)
                        r = x + y if op == "add" else x - y
                        return {"frame_num": r.frame_num, "seconds": r.seconds, "pts": r.pts,
                                "time_base": [r.time_base.numerator, r.time_base.denominator], "repr": repr(r)}
                    out["pts_arith"].append({"a": a, "b": b, "fps": f, "op": op, **outcome(run)})
                def cmp():
                    x, y = mk(a, f), mk(b, f)
                    return {"eq": x == y, "ne": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}
                out["pts_compare"].append({"a": a, "b": b, "fps": f, **outcome(cmp)})
    # two PTS-backed values of different nominal rates compare by frame number, not exactly
    def cross_pts():
        x = FrameTimecode(Timecode(1001, Fraction(1, 30000)), 25.0)
        y = FrameTimecode(Timecode(1001, Fraction(1, 30000)), 30.0)
        return {"eq": x == y, "lt": x < y, "hash_x": hash(x), "hash_y": hash(y)}
    out["pts_cross_rate"] = outcome(cross_pts)
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "timecode_cases.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
