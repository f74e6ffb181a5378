"""Differential fuzz of the plug-in SURFACE: detector constructors with arguments of every kind (valid, out of range, wrong type),
``SceneManager`` property setters / getters, and the module-level helpers, ``pyscenedetect_amd`` against the unmodified reference.
Build container only; no GPU.

    python tools/fuzz_api_vs_reference.py [--seconds 60] [--seed 1]

Outcome per case: the exception TYPE, or what a caller can observe -- ``get_metrics()``, ``event_buffer_length``, the public
attributes both classes have, the manager's properties after a random sequence of assignments, the return value of
``compute_downscale_factor`` / ``get_scenes_from_cuts`` on random inputs."""
import argparse
import json
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402

import scenedetect as ref  # noqa: E402
from scenedetect import scene_manager as ref_sm  # noqa: E402
from scenedetect.common import Interpolation as RefInterp  # noqa: E402
from scenedetect.detector import FlashFilter as RefFF  # noqa: E402
from scenedetect.detectors import (AdaptiveDetector as RA, ContentDetector as RC, HashDetector as RH, HistogramDetector as RHi,  # noqa: E402
                                   ThresholdDetector as RT)

import pyscenedetect_amd as psd  # noqa: E402
from pyscenedetect_amd import scene_manager as our_sm  # noqa: E402

REF = {"ContentDetector": RC, "AdaptiveDetector": RA, "HistogramDetector": RHi, "ThresholdDetector": RT, "HashDetector": RH}
PARAMS = {
    "ContentDetector": ["threshold", "min_scene_len", "weights", "luma_only", "kernel_size", "filter_mode"],
    "AdaptiveDetector": ["adaptive_threshold", "min_scene_len", "window_width", "min_content_val", "weights", "luma_only", "kernel_size"],
    "HistogramDetector": ["threshold", "bins", "min_scene_len"],
    "ThresholdDetector": ["threshold", "min_scene_len", "fade_bias", "add_final_scene", "method"],
    "HashDetector": ["threshold", "size", "lowpass", "min_scene_len"],
}
ODD = [None, -1, 0, 1, 2, 3, 4, 7, 16, 255, 256, 1000, -0.5, 0.0, 0.5, 1.0, 2.5, 27.0, 1e9, "3", "0.5", "1.5s", "00:00:01.000", "abc", "", True, False, [1, 2], (1, 1, 1, 1),
       np.int64(3), np.int32(15), np.uint8(7), np.float64(2.5), np.float32(0.5), np.bool_(True)]      # (numbers that came out of an array)


def draw_arg(rng, name):
    if rng.integers(0, 4) == 0:
        v = ODD[int(rng.integers(0, len(ODD)))]
        return v
    if name in ("threshold", "adaptive_threshold", "min_content_val"):
        return float(round(rng.uniform(-1, 100), 2)) if rng.integers(0, 2) else int(rng.integers(0, 100))
    if name == "min_scene_len":
        return [int(rng.integers(0, 50)), float(round(rng.uniform(0, 2), 2)), "%.2fs" % rng.uniform(0, 2), "00:00:%06.3f" % rng.uniform(0, 3), str(int(rng.integers(0, 50)))][int(rng.integers(0, 5))]
    if name == "weights":
        return [float(rng.integers(0, 3)) for _ in range(4)]
    if name in ("luma_only", "add_final_scene"):
        return bool(rng.integers(0, 2))
    if name == "kernel_size":
        return int(rng.integers(-3, 12))
    if name == "filter_mode":
        return ["MERGE", "SUPPRESS"][int(rng.integers(0, 2))]
    if name == "window_width":
        return int(rng.integers(-1, 6))
    if name == "bins":
        return int(rng.choice([0, 1, 2, 16, 100, 128, 256, 257, 512, -4]))
    if name == "fade_bias":
        return float(round(rng.uniform(-1.5, 1.5), 2))
    if name == "method":
        return ["FLOOR", "CEILING"][int(rng.integers(0, 2))]
    if name in ("size", "lowpass"):
        return int(rng.integers(-1, 20))
    return None


def build(side, name, kw):
    kw = dict(kw)
    if side == "ref":
        cls, comp, ff, meth = REF[name], RC.Components, RFF_MODE, RT.Method
    else:
        cls, comp, ff, meth = getattr(psd, name), psd.ContentDetector.Components, psd.FlashFilter.Mode, psd.ThresholdDetector.Method
    if isinstance(kw.get("weights"), list) and len(kw["weights"]) == 4:
        kw["weights"] = comp(*kw["weights"])
    if kw.get("filter_mode") in ("MERGE", "SUPPRESS"):
        kw["filter_mode"] = ff[kw["filter_mode"]]
    if kw.get("method") in ("FLOOR", "CEILING"):
        kw["method"] = meth[kw["method"]]
    return cls(**kw)


RFF_MODE = RFF = RefFF.Mode


def observe(det):
    out = {"metrics": list(det.get_metrics()), "event_buffer_length": int(det.event_buffer_length)}
    for attr in ("threshold", "adaptive_threshold", "min_content_val", "window_width", "bins", "fade_bias", "add_final_scene"):
        if hasattr(det, attr):
            v = getattr(det, attr)
            out[attr] = v if isinstance(v, (int, float, bool, str, type(None))) else str(v)
    return out


def ctor_case(rng):
    name = list(PARAMS)[int(rng.integers(0, 5))]
    kw = {}
    for p in PARAMS[name]:
        if rng.integers(0, 3) == 0:
            kw[p] = draw_arg(rng, p)
    return ("ctor", name, kw)


def manager_case(rng):
    prog = []
    for _ in range(int(rng.integers(1, 7))):
        k = int(rng.integers(0, 6))
        if k == 0:
            prog.append(("downscale", ODD[int(rng.integers(0, len(ODD)))] if rng.integers(0, 3) == 0 else int(rng.integers(-1, 6))))
        elif k == 1:
            prog.append(("auto_downscale", bool(rng.integers(0, 2))))
        elif k == 2:
            # (a value that is not an Interpolation is refused at the assignment where the reference fails later, at its first use: a
            #  documented difference, not drawn.  CUBIC, accepted since round 6, takes every other LANCZOS4 draw so that the cases of
            #  earlier seeds stay what they were)
            name = ["NEAREST", "LINEAR", "AREA", "LANCZOS4"][int(rng.integers(0, 4))]
            prog.append(("interpolation", "CUBIC" if name == "LANCZOS4" and len(prog) % 2 else name))
        elif k == 3:
            c = [(0, 0, 10, 10), (5, 5, 2, 2), (-1, 0, 3, 3), (0, 0, 0, 0), (1, 2, 3), None, "crop", (1.5, 0, 3, 3), (3, 4, 100000, 7)][int(rng.integers(0, 9))]
            prog.append(("crop", c))
        elif k == 4:
            prog.append(("read",))
        else:
            prog.append(("num_detectors",))
    return ("manager", prog)


def helper_case(rng):
    k = int(rng.integers(0, 2))
    if k == 0:
        return ("downscale_factor", int(rng.integers(1, 9000)), int(rng.choice([256, 128, 1, 1000])))
    n = int(rng.integers(0, 6))
    cuts = sorted({int(x) for x in rng.integers(1, 200, n)})
    return ("scenes_from_cuts", cuts, int(rng.integers(0, 3)), int(rng.integers(3, 260)), float(rng.choice([25.0, 29.97])))


def run(side, case):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if case[0] == "ctor":
            return observe(build(side, case[1], case[2]))
        if case[0] == "manager":
            sm = ref.SceneManager() if side == "ref" else psd.SceneManager(engine=object())
            interp = RefInterp if side == "ref" else psd.Interpolation
            out = []
            for step in case[1]:
                try:
                    if step[0] == "downscale":
                        sm.downscale = step[1]
                        out.append(["ok"])
                    elif step[0] == "auto_downscale":
                        sm.auto_downscale = step[1]
                        out.append(["ok"])
                    elif step[0] == "interpolation":
                        sm.interpolation = interp[step[1]] if step[1] != "bad" else "bad"
                        out.append(["ok"])
                    elif step[0] == "crop":
                        sm.crop = step[1]
                        out.append(["ok"])
                    elif step[0] == "read":
                        out.append(["ok", sm.downscale, sm.auto_downscale, sm.interpolation.name, None if sm.crop is None else list(sm.crop)])
                    else:
                        out.append(["ok", sm.get_num_detectors()])
                except Exception as ex:  # noqa: BLE001
                    out.append(["raises", type(ex).__name__, "" if isinstance(ex, AssertionError) else str(ex)])
            return out
        if case[0] == "downscale_factor":
            f = ref_sm.compute_downscale_factor if side == "ref" else our_sm.compute_downscale_factor
            return f(case[1], case[2])
        f = ref_sm.get_scenes_from_cuts if side == "ref" else our_sm.get_scenes_from_cuts
        TC = ref.FrameTimecode if side == "ref" else psd.FrameTimecode
        fps = case[4]
        scenes = f([TC(c, fps) for c in case[1]], TC(case[2], fps), TC(case[3], fps))
        return [[a.frame_num, b.frame_num] for a, b in scenes]


def outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001
        return {"raises": type(ex).__name__, "message": str(ex)}


def draw_case(rng):
    k = int(rng.integers(0, 10))
    if k < 6:
        return ctor_case(rng)
    if k < 9:
        return manager_case(rng)
    return helper_case(rng)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    t_end = time.time() + args.seconds
    cases, raised, bad = 0, 0, []
    while time.time() < t_end:
        case = draw_case(np.random.default_rng([args.seed, cases]))
        a, b = outcome(lambda: run("ref", case)), outcome(lambda: run("ours", case))
        raised += "raises" in a
        if a != b:
            desc = {"case": cases, "spec": case, "ref": str(a)[:300], "ours": str(b)[:300]}
            bad.append(desc)
            if args.verbose:
                print(json.dumps(desc, default=str), flush=True)
            if len(bad) >= 40:
                break
        cases += 1
    print(json.dumps({"seed": args.seed, "cases": cases, "cases_that_raise_on_both_sides": raised, "mismatches": bad[:40]}, default=str))


if __name__ == "__main__":
    main()
