"""Differential fuzz of the NATIVE whole-clip decision epilogues (``psd_epilogue_*`` through ``pyscenedetect_amd.corpus.decide``: what
``detect_corpus`` and ``bench.py`` decide with) against the unmodified reference's own ``SceneManager`` run over the same clip.
Build container only; records come from the CPU oracle; no GPU.

    python tools/fuzz_epilogue_vs_reference.py [--seconds 60] [--seed 1]

Every case: a random synthetic clip, a frame rate, one detector of the five the epilogues replay (content / adaptive / hist / hash /
threshold) with random constructor arguments (``min_scene_len`` as frames / seconds / strings, both filter modes, edge weights, bins,
fade bias, FLOOR / CEILING, ``add_final_scene``), the reference run WITHOUT a downscale -- and the cut list must be the same.

``--downscale`` (round 6): the reference run WITH the resize its SceneManager puts in front of the detectors -- its default
``auto_downscale`` or a manual factor, each interpolation the engine implements -- against ``corpus.detect_corpus`` over the oracle engine
with the same setting: the packed / sharded flow computes what ``detect(video, detector)`` computes (every fourth clip is wider than 256
pixels, where the default pipeline resizes)."""
import argparse
import json
import logging
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT, "/root/reference"]
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import fuzz_host_vs_reference as F  # noqa: E402  (clips, the reference-side runner)
from oracle.detectors_np import score_batch as oracle_score  # noqa: E402
from pyscenedetect_amd import corpus  # noqa: E402

NAMES = {"content": "ContentDetector", "adaptive": "AdaptiveDetector", "hist": "HistogramDetector", "threshold": "ThresholdDetector",
         "hash": "HashDetector"}


WIDE = False


def draw(rng):
    frames = F.draw_clip(rng)
    fps = [25.0, 30.0, 24.0, 29.97, 23.976, 60.0, 12.5][int(rng.integers(0, 7))]
    name = list(NAMES)[int(rng.integers(0, 5))]
    kw = {}
    if rng.integers(0, 4):
        kw["min_scene_len"] = F.draw_min_scene_len(rng)
    kernel = 0
    if name in ("content", "adaptive"):
        if rng.integers(0, 2):
            w = [float(rng.integers(0, 3)) for _ in range(3)] + [float(rng.integers(0, 2))]
            if WIDE and rng.integers(0, 2):      # fractional and negative weights (the score divides by the sum of their magnitudes)
                w = [float(round(rng.uniform(-1.5, 2.5), 2)) if rng.integers(0, 4) else 0.0 for _ in range(4)]
            if sum(abs(x) for x in w) == 0 and not (WIDE and rng.integers(0, 2)):      # (all zero: NaN scores, no cut, no exception)
                w[2] = 1.0
            kw["weights"] = w
        if rng.integers(0, 3) == 0:
            kernel = int(rng.choice([3, 5, 7]))
    if name == "content":
        if rng.integers(0, 3):
            kw["threshold"] = float(round(rng.uniform(4.0, 70.0), 2))
        if rng.integers(0, 2):
            kw["filter_mode"] = int(rng.integers(0, 2))
    elif name == "adaptive":
        if rng.integers(0, 2):
            kw["adaptive_threshold"] = float(round(rng.uniform(1.2, 6.0), 2))
        if rng.integers(0, 2):
            kw["window_width"] = int(rng.integers(1, 5))
        if rng.integers(0, 2):
            kw["min_content_val"] = float(round(rng.uniform(2.0, 30.0), 2))
    elif name == "hist":
        if rng.integers(0, 2):
            kw["threshold"] = float(round(rng.uniform(0.01, 0.6), 3))
        if rng.integers(0, 2):
            kw["bins"] = int(rng.choice([16, 32, 64, 100, 128, 200, 256]))
    elif name == "hash":
        if rng.integers(0, 2):
            kw["threshold"] = float(round(rng.uniform(0.1, 0.6), 3))
        if rng.integers(0, 2):
            kw["size"] = int(rng.choice([2, 3, 4, 8, 12, 16, 32] if WIDE else [8, 16]))
        if rng.integers(0, 2):
            kw["lowpass"] = int(rng.choice([1, 2, 3, 4, 5] if WIDE else [1, 2, 4]))
    else:
        if rng.integers(0, 2):
            kw["threshold"] = int(rng.integers(3, 120))
            if WIDE and rng.integers(0, 3) == 0:
                kw["threshold"] = float(round(rng.uniform(0.0, 140.0), 2))     # the reference truncates it to an int
        if rng.integers(0, 2):
            kw["fade_bias"] = float(round(rng.uniform(-1.0, 1.0), 2))
        if rng.integers(0, 2):
            kw["add_final_scene"] = True
        if rng.integers(0, 3) == 0:
            kw["method"] = int(rng.integers(0, 2))
    return frames, fps, name, kw, kernel


def draw_downscale(rng):
    """SceneManager's resize settings: {"auto_downscale": bool[, "downscale": int][, "interpolation": name]}"""
    cfg = {"auto_downscale": bool(rng.integers(0, 3))}
    if not cfg["auto_downscale"]:
        cfg["downscale"] = int(rng.integers(1, 5))
    if rng.integers(0, 3) == 0:
        cfg["interpolation"] = ["LINEAR", "NEAREST", "AREA"][int(rng.integers(0, 3))]
    return cfg


def reference_cuts(frames, fps, name, kw, kernel, resize=None):
    ref_kw = dict(kw)
    if "filter_mode" in ref_kw:
        ref_kw["filter_mode"] = ["MERGE", "SUPPRESS"][ref_kw["filter_mode"]]
    if "method" in ref_kw:
        ref_kw["method"] = ["FLOOR", "CEILING"][ref_kw["method"]]
    if kernel:
        ref_kw["kernel_size"] = kernel
    cfg = {"stats": False, "auto_downscale": False, "start_in_scene": False}
    cfg.update(resize or {})
    return F.run_side("ref", frames, fps, [(NAMES[name], ref_kw)], cfg, None)["cuts"]


def corpus_cuts(frames, fps, name, kw, kernel, resize):
    """The same decision from the packed flow's entry point over the oracle engine (``corpus.detect_corpus``: records of the resized
    frames -> ``decide`` with the resized size)."""
    from oracle.detectors_np import OracleEngine

    params = dict(kw)
    if "weights" in params:
        params["weights"] = tuple(params["weights"])
    interp = {"NEAREST": 0, "LINEAR": 1, "AREA": 3}[resize.get("interpolation", "LINEAR")]
    res = corpus.detect_corpus(OracleEngine(), [frames], fps, {name: params}, edge_kernel=kernel, auto_downscale=resize["auto_downscale"],
                               downscale=resize.get("downscale", 1), interpolation=interp)
    return [int(c) for c in res[0][name]]


def native_cuts(frames, fps, name, kw, kernel):
    if name == "hash":      # HashDetector: thumbnails (oracle) -> native DCT / median bits -> native decisions
        from oracle import lib as orc
        from pyscenedetect_amd import epilogue

        size, lowpass = kw.get("size", 8), kw.get("lowpass", 2)
        bits = epilogue.hash_bits(orc.hash_thumbs(frames, size * lowpass), size)
        return [int(c) for c in epilogue.hash_cuts(bits, fps, kw.get("threshold", 0.35), kw.get("min_scene_len", 15))[0]]
    w = kw.get("weights")
    edges = name in ("content", "adaptive") and w is not None and w[3] > 0.0
    h, wd = frames.shape[1:3]
    if edges and not kernel:
        from pyscenedetect_amd.detectors.content_detector import estimated_kernel_size

        kernel = estimated_kernel_size(wd, h)
    recs = oracle_score(frames, edges=edges, kernel_size=kernel)
    params = dict(kw)
    if "weights" in params:
        params["weights"] = tuple(params["weights"])
    return [int(c) for c in corpus.decide(recs, h, wd, fps, {name: params})[name]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--wide", action="store_true", help="fractional / negative weights, fractional fade thresholds, odd hash sizes")
    ap.add_argument("--tiny", action="store_true", help="frames of a few pixels, clips of hundreds of frames")
    ap.add_argument("--downscale", action="store_true", help="the reference behind its resize (auto / manual, each interpolation) against detect_corpus")
    args = ap.parse_args()
    global WIDE
    WIDE, F.TINY = args.wide, args.tiny
    logging.disable(logging.CRITICAL)
    warnings.simplefilter("ignore")
    t_end = time.time() + args.seconds
    cases, by, bad = 0, {}, []
    while time.time() < t_end:
        rng = np.random.default_rng([args.seed, cases])
        frames, fps, name, kw, kernel = draw(rng)
        resize = None
        if args.downscale:
            while name == "hash":      # (HashDetector is not one of the corpus flow's detectors)
                frames, fps, name, kw, kernel = draw(rng)
            resize = draw_downscale(rng)
            a = F.decisions(F.outcome(lambda: {"cuts": reference_cuts(frames, fps, name, kw, kernel, resize)}))
            b = F.decisions(F.outcome(lambda: {"cuts": corpus_cuts(frames, fps, name, kw, kernel, resize)}))
        else:
            a = F.decisions(F.outcome(lambda: {"cuts": reference_cuts(frames, fps, name, kw, kernel)}))
            b = F.decisions(F.outcome(lambda: {"cuts": native_cuts(frames, fps, name, kw, kernel)}))
        by[name] = by.get(name, 0) + 1
        if a != b:
            desc = {"case": cases, "resize": resize, "shape": list(frames.shape), "fps": fps, "detector": name, "params": kw, "kernel": kernel, "ref": str(a)[:200], "native": str(b)[:200]}
            bad.append(desc)
            if args.verbose:
                print(json.dumps(desc), flush=True)
            if len(bad) >= 25:
                break
        cases += 1
    print(json.dumps({"seed": args.seed, "cases": cases, "by_detector": by, "mismatches": bad[:25]}))


if __name__ == "__main__":
    main()
