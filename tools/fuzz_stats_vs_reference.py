"""Differential fuzz of ``pyscenedetect_amd.StatsManager`` against the reference's (``/root/reference/scenedetect/stats_manager.py``).
Build container only; no GPU.

    python tools/fuzz_stats_vs_reference.py [--seconds 60] [--seed 1]

A case is a random program on one manager per side: ``register_metrics``, ``set_metrics`` / ``get_metrics`` / ``metrics_exist`` with
frame numbers or FrameTimecodes (frame-, seconds- and timestamp-backed), ``is_save_required``, ``metric_keys``, ``save_to_csv`` to a
path / pathlib.Path / open file with and without ``force_save`` and a base timecode, ``load_from_csv`` of what was saved, of what the
OTHER side saved, and of damaged files (missing header rows, wrong column names, non-numeric cells, extra / missing cells, empty
file), ``valid_header``.  Same return values, same file text, same exception types."""
import argparse
import io
import json
import os
import pathlib
import sys
import tempfile
import time
import warnings
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402

import scenedetect as ref  # noqa: E402
from scenedetect.common import Timecode as RefT  # noqa: E402
from scenedetect.stats_manager import StatsManager as RefStats  # noqa: E402

import pyscenedetect_amd as psd  # noqa: E402

KEYS = ["content_val", "delta_hue", "delta_sat", "delta_lum", "delta_edges", "hist_diff [bins=128]", "average_rgb", "adaptive_ratio (w=2)",
        "hash_dist [size=16 lowpass=2]", "my metric", "x"]


def tc(side, spec, fps):
    kind, v = spec
    TC = ref.FrameTimecode if side == "ref" else psd.FrameTimecode
    T = RefT if side == "ref" else psd.Timecode
    if kind == "int":
        return v
    if kind == "tc_frames":
        return TC(v, fps)
    if kind == "tc_secs":
        return TC(float(v), fps)
    return TC(T(v, Fraction(1, 1000)), fps)


def draw_tc(rng):
    k = int(rng.integers(0, 6))
    if k <= 2:
        return ("int", int(rng.integers(0, 40)))
    if k == 3:
        return ("tc_frames", int(rng.integers(0, 40)))
    if k == 4:
        return ("tc_secs", float(rng.integers(0, 40)) / 25.0)
    return ("tc_pts", int(rng.integers(0, 40)) * 40)


def draw_value(rng):
    k = int(rng.integers(0, 8))
    if k == 0:
        return int(rng.integers(0, 1000))
    if k == 1:
        return None
    if k == 2:
        return float(rng.choice([0.0, 1.0, 255.0, 1e-9, 1e12, 0.1 + 0.2]))
    return float(rng.uniform(0, 255))


def damage(text, rng):
    lines = text.split("\n")
    k = int(rng.integers(0, 9))
    if k == 0:
        return ""
    if k == 1 and len(lines) > 1:
        return "\n".join(lines[1:])
    if k == 2:
        return text.replace("Frame Number", "Frame", 1)
    if k == 3:
        return text.replace("Timecode", "Time", 1)
    if k == 4 and len(lines) > 2:
        lines[2] = lines[2] + ",7.5"
        return "\n".join(lines)
    if k == 5 and len(lines) > 2:
        lines[2] = ",".join(lines[2].split(",")[:-1])
        return "\n".join(lines)
    if k == 6 and len(lines) > 2:
        cells = lines[2].split(",")
        cells[-1] = "abc"
        lines[2] = ",".join(cells)
        return "\n".join(lines)
    if k == 7 and len(lines) > 2:
        cells = lines[2].split(",")
        cells[0] = "x1"
        lines[2] = ",".join(cells)
        return "\n".join(lines)
    return text.replace(",", ";")


def draw_program(rng):
    fps = float(rng.choice([25.0, 29.97, 24.0, 60.0]))
    prog = [("new", bool(rng.integers(0, 3) == 0))]
    for _ in range(int(rng.integers(2, 14))):
        k = int(rng.integers(0, 12))
        if k == 0:
            prog.append(("register", [KEYS[int(i)] for i in rng.integers(0, len(KEYS), int(rng.integers(1, 4)))]))
        elif k <= 3:
            prog.append(("set", draw_tc(rng), {KEYS[int(i)]: draw_value(rng) for i in rng.integers(0, len(KEYS), int(rng.integers(1, 4)))}))
        elif k == 4:
            prog.append(("get", draw_tc(rng), [KEYS[int(i)] for i in rng.integers(0, len(KEYS), int(rng.integers(1, 3)))]))
        elif k == 5:
            prog.append(("exist", draw_tc(rng), [KEYS[int(i)] for i in rng.integers(0, len(KEYS), int(rng.integers(1, 3)))]))
        elif k == 6:
            prog.append(("save_required",))
        elif k == 7:
            prog.append(("keys",))
        elif k == 8:
            prog.append(("save", ["str", "path", "file"][int(rng.integers(0, 3))], bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
        elif k == 9:
            prog.append(("load_own", ["str", "path", "file"][int(rng.integers(0, 3))]))
        elif k == 10:
            prog.append(("load_damaged", int(rng.integers(0, 1 << 30))))
        else:
            prog.append(("valid_header", [["Frame Number", "Timecode", "a"], ["Frame Number", "Timecode"], ["Timecode", "Frame Number", "a"], ["Frame Number"], [],
                                          ["Frame Number", "Timecode", "a", "a"]][int(rng.integers(0, 6))]))
    return fps, prog


def run(side, fps, prog, tmp, other_text=None):
    Stats = RefStats if side == "ref" else psd.StatsManager
    TC = ref.FrameTimecode if side == "ref" else psd.FrameTimecode
    out, sm, saved = [], None, None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in prog:
            op = step[0]
            try:
                if op == "new":
                    sm = Stats(TC(0, fps)) if step[1] else Stats()
                    res = None
                elif op == "register":
                    res = sm.register_metrics(step[1])
                elif op == "set":
                    res = sm.set_metrics(tc(side, step[1], fps), dict(step[2]))
                elif op == "get":
                    res = sm.get_metrics(tc(side, step[1], fps), step[2])
                elif op == "exist":
                    res = sm.metrics_exist(tc(side, step[1], fps), step[2])
                elif op == "save_required":
                    res = sm.is_save_required()
                elif op == "keys":
                    res = sorted(sm.metric_keys)
                elif op == "save":
                    path = os.path.join(tmp, side + ".csv")
                    kw = {"force_save": step[2]}
                    if step[1] == "file":
                        with open(path, "w", newline="") as f:
                            sm.save_to_csv(f, **kw)
                    else:
                        sm.save_to_csv(pathlib.Path(path) if step[1] == "path" else path, **kw)
                    saved = open(path).read() if os.path.exists(path) else None
                    res = saved
                elif op == "load_own":
                    path = os.path.join(tmp, side + "_in.csv")
                    text = saved if saved is not None else "Frame Number,Timecode,content_val\n1,00:00:00.000,1.5\n2,00:00:00.040,2.5\n"
                    open(path, "w").write(text)
                    if step[1] == "file":
                        with open(path) as f:
                            res = sm.load_from_csv(f)
                    else:
                        res = sm.load_from_csv(pathlib.Path(path) if step[1] == "path" else path)
                elif op == "load_damaged":
                    path = os.path.join(tmp, side + "_bad.csv")
                    base = saved if saved is not None else "Frame Number,Timecode,content_val,x\n1,00:00:00.000,1.5,2\n2,00:00:00.040,2.5,3\n"
                    open(path, "w").write(damage(base, np.random.default_rng(step[1])))
                    res = sm.load_from_csv(path)
                else:
                    res = Stats.valid_header(step[1])
                out.append(["ok", res])
            except Exception as ex:  # noqa: BLE001 -- the type is the outcome
                out.append(["raises", type(ex).__name__, "" if isinstance(ex, AssertionError) else str(ex)])
    return out


def logged(fn):
    """The program's results followed by what it logged (level and text; temporary paths blanked) and the warnings it emitted."""
    import logging
    import re
    import warnings

    class Capture(logging.Handler):
        def __init__(self):
            super().__init__(logging.DEBUG)
            self.records = []

        def emit(self, record):
            self.records.append([record.levelname, re.sub(r"/tmp/\S+", "<tmp>", record.getMessage())])

    log, capture = logging.getLogger("pyscenedetect"), Capture()
    saved = (log.level, log.propagate, logging.root.manager.disable)
    log.addHandler(capture)
    log.setLevel(logging.DEBUG)
    log.propagate = False
    logging.disable(logging.NOTSET)
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            out = list(fn())
    finally:
        log.removeHandler(capture)
        log.setLevel(saved[0])
        log.propagate = saved[1]
        logging.disable(saved[2])
    out.append(["log", capture.records])
    out.append(["warnings", sorted({(w.category.__name__, re.sub(r"/tmp/\S+", "<tmp>", str(w.message))) for w in caught
                                    if not issubclass(w.category, ResourceWarning)})])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    t_end = time.time() + args.seconds
    cases, bad = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        while time.time() < t_end:
            rng = np.random.default_rng([args.seed, cases])
            fps, prog = draw_program(rng)
            a, b = logged(lambda: run("ref", fps, prog, tmp)), logged(lambda: run("ours", fps, prog, tmp))
            if a != b:
                first = next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)
                desc = {"case": cases, "fps": fps, "step": first, "op": prog[first], "ref": str(a[first])[:300], "ours": str(b[first])[:300], "program": prog[:first + 1]}
                bad.append(desc)
                if args.verbose:
                    print(json.dumps(desc, default=str), flush=True)
                if len(bad) >= 25:
                    break
            cases += 1
    print(json.dumps({"seed": args.seed, "cases": cases, "mismatches": bad[:25]}, default=str))


if __name__ == "__main__":
    main()
