"""Differential fuzz of ``pyscenedetect_amd.timecode`` (FrameTimecode / Timecode / framerate_to_fraction) against the unmodified
reference classes (``/root/reference/scenedetect/common.py`` over the cv2 shim).  Build container only; no GPU.

    python tools/fuzz_timecode_vs_reference.py [--seconds 60] [--seed 1]

Every case builds one or two timecodes on both sides from random ingredients -- frame numbers, seconds (round and awkward floats),
timecode strings in every accepted spelling and a few rejected ones, presentation timestamps with random time bases, frame rates
as floats / Fractions / NTSC pairs -- and applies a random short program: ``+`` / ``-`` with another timecode or a plain int /
float / str, the six comparisons, ``get_timecode`` with a random precision and rounding, ``frame_num`` / ``seconds`` / ``hash`` /
``int`` / ``float`` / ``str`` / ``repr``-free views, ``equal_frame_rate``.  The two sides must produce the same values (floats bit for
bit) or raise the same exception type.  Prints one JSON line with the first mismatches."""
import argparse
import json
import os
import sys
import time
import warnings
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402

from scenedetect.common import FrameTimecode as RefTC, Timecode as RefT, framerate_to_fraction as ref_rate  # noqa: E402

from pyscenedetect_amd.timecode import FrameTimecode as OurTC, Timecode as OurT, framerate_to_fraction as our_rate  # noqa: E402


def draw_fps(rng):
    k = int(rng.integers(0, 10))
    if k == 0:
        return ("frac", [24000, 1001])
    if k == 1:
        return ("frac", [30000, 1001])
    if k == 2:
        return ("frac", [int(rng.integers(1, 240000)), int(rng.integers(1, 2000))])
    if k == 3:
        return ("float", float(round(rng.uniform(0.5, 240.0), int(rng.integers(0, 5)))))
    return ("float", float(rng.choice([1.0, 10.0, 12.5, 23.976, 24.0, 25.0, 29.97, 30.0, 50.0, 59.94, 60.0, 120.0])))


def fps_of(side, spec):
    kind, v = spec
    return Fraction(v[0], v[1]) if kind == "frac" else v


def draw_value(rng):
    k = int(rng.integers(0, 16))
    if k == 0:
        return ("int", int(rng.integers(0, 200000)))
    if k == 1:
        return ("int", int(rng.integers(0, 50)))
    if k == 2:
        return ("float", float(rng.uniform(0, 7200)))
    if k == 3:
        return ("float", float(round(rng.uniform(0, 100), int(rng.integers(0, 4)))))
    if k == 4:
        h, m, s = int(rng.integers(0, 30)), int(rng.integers(0, 60)), rng.uniform(0, 60)
        return ("str", "%02d:%02d:%06.3f" % (h, m, s))
    if k == 5:
        return ("str", "%d:%02d:%02d" % (int(rng.integers(0, 5)), int(rng.integers(0, 60)), int(rng.integers(0, 60))))
    if k == 6:
        return ("str", "%.3fs" % rng.uniform(0, 500))
    if k == 7:
        return ("str", str(int(rng.integers(0, 100000))))
    if k == 8:
        return ("str", "%.2f" % rng.uniform(0, 500))
    if k == 9:
        return ("str", "%02d:%06.3f" % (int(rng.integers(0, 60)), rng.uniform(0, 60)))
    if k == 10:
        return ("pts", [int(rng.integers(0, 10 ** 7)), [1, int(rng.choice([1000, 90000, 12800, 30000, 1001, 600, 48000]))]])
    if k == 11:
        return ("bad", rng.choice(["abc", "1:2:3:4", "00:61:00", "00:00:61", "-5", "5x", "", "1.5.5", "12:", ":30", "1e3", " 12 ", "1,5"]).item())
    if k == 12:
        return ("int", -int(rng.integers(1, 50)))
    if k == 13:
        return ("float", -float(rng.uniform(0.01, 5)))
    if k == 14:
        return ("none", None) if rng.integers(0, 3) else ("numpy", [["int64", "int32", "int64", "float64", "intp"][int(rng.integers(0, 5))],
                                                                     int(rng.integers(0, 200))])
    return ("float", float(rng.integers(0, 1000)) / float(rng.choice([1, 2, 4, 8, 25, 30, 1000])))


def make(side, value, fps_spec):
    kind, v = value
    TC, T = (RefTC, RefT) if side == "ref" else (OurTC, OurT)
    if kind == "pts":
        return TC(T(v[0], Fraction(v[1][0], v[1][1])), fps_of(side, fps_spec))
    if kind == "numpy":          # a frame number that came out of an array
        v = getattr(np, v[0])(v[1])
    return TC(v, fps_of(side, fps_spec))


def operand(side, spec, fps_spec):
    """A right-hand side: a timecode of the same side, or a plain value."""
    kind, v = spec
    if kind == "tc":
        return make(side, v[0], v[1])
    if kind == "numpy":
        return getattr(np, v[0])(v[1])
    return v


def views(tc, precision, rounding):
    out = {}
    for name, fn in (("frame_num", lambda: tc.frame_num), ("seconds", lambda: tc.seconds), ("timecode", lambda: tc.get_timecode(precision, rounding)),
                     ("hash", lambda: hash(tc)), ("int", lambda: int(tc)), ("float", lambda: float(tc)), ("str", lambda: str(tc)),
                     ("frame_rate", lambda: Fraction(tc.frame_rate)), ("pts", lambda: tc.pts), ("time_base", lambda: Fraction(tc.time_base))):
        try:
            out[name] = fn()
        except Exception as ex:  # noqa: BLE001
            out[name] = "raises:" + type(ex).__name__ + ":" + str(ex)
    return out


def run(side, case):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = make(side, case["a"], case["fps"])
        res = [views(a, case["precision"], case["rounding"])]
        for op, rhs_spec in case["program"]:
            rhs = operand(side, rhs_spec, case["fps2"])
            if op == "+":
                a = a + rhs
                res.append(views(a, case["precision"], case["rounding"]))
            elif op == "-":
                a = a - rhs
                res.append(views(a, case["precision"], case["rounding"]))
            elif op == "cmp":
                row = []
                for f in (lambda: a == rhs, lambda: a != rhs, lambda: a < rhs, lambda: a <= rhs, lambda: a > rhs, lambda: a >= rhs):
                    try:
                        row.append(bool(f()))
                    except Exception as ex:  # noqa: BLE001
                        row.append("raises:" + type(ex).__name__)
                res.append(row)
            elif op == "same_rate":
                res.append(bool(a.equal_frame_rate(rhs)) if hasattr(a, "equal_frame_rate") else None)
        return res


def outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001
        return {"raises": type(ex).__name__, "message": str(ex)}


def draw_case(rng):
    fps = draw_fps(rng)
    fps2 = fps if rng.integers(0, 3) else draw_fps(rng)
    prog = []
    for _ in range(int(rng.integers(0, 4))):
        op = ["+", "-", "cmp", "cmp", "same_rate"][int(rng.integers(0, 5))]
        if op == "same_rate" or rng.integers(0, 2):
            rhs = ("tc", [draw_value(rng), fps2])
        else:
            drawn = draw_value(rng)
            rhs = ("numpy", drawn[1]) if drawn[0] == "numpy" else ("plain", drawn[1])
            if rhs[0] == "plain" and isinstance(rhs[1], list):
                rhs = ("plain", 3)
        prog.append((op, rhs))
    return {"a": draw_value(rng), "fps": fps, "fps2": fps2, "program": prog, "precision": int(rng.integers(0, 7)), "rounding": bool(rng.integers(0, 2))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    t_end = time.time() + args.seconds
    cases, raised, bad = 0, 0, []
    while time.time() < t_end:
        rng = np.random.default_rng([args.seed, cases])
        case = draw_case(rng)
        a = outcome(lambda: run("ref", case))
        b = outcome(lambda: run("ours", case))
        raised += "raises" in a
        if a != b:
            desc = {"case": cases, "spec": case, "ref": str(a)[:400], "ours": str(b)[:400]}
            bad.append(desc)
            if args.verbose:
                print(json.dumps(desc, default=str), flush=True)
            if len(bad) >= 25:
                break
        cases += 1
    # frame-rate parsing on its own
    rate_bad = []
    rng = np.random.default_rng([args.seed, 10 ** 9])
    for _ in range(20000):
        f = float(round(rng.uniform(0.01, 300.0), int(rng.integers(0, 6)))) if rng.integers(0, 4) else float(rng.choice([23.976, 29.97, 59.94, 119.88, 23.98, 29.970029, 24.0, 25.0]))
        x, y = outcome(lambda: Fraction(ref_rate(f))), outcome(lambda: Fraction(our_rate(f)))
        if x != y:
            rate_bad.append([f, str(x), str(y)])
    print(json.dumps({"seed": args.seed, "cases": cases, "cases_that_raise_on_both_sides": raised, "mismatches": bad[:25],
                      "framerate_to_fraction_mismatches": rate_bad[:10]}, default=str))


if __name__ == "__main__":
    main()
