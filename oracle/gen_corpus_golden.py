"""Generate tests/golden/corpus_default_pipeline.json: cut lists of the UNMODIFIED reference's DEFAULT pipeline.

Run in the build container only (needs /root/reference):

    python oracle/gen_corpus_golden.py

For every seeded clip below the reference's own ``SceneManager`` is run exactly as ``scenedetect.detect(video, detector)`` runs it
(``scenedetect/__init__.py:203-216``: no StatsManager, ``auto_downscale=True`` -- every frame goes through
``cv2.resize(frame, (round(w / f), round(h / f)), INTER_LINEAR)`` with ``f = compute_downscale_factor(max(frame_size))``,
``scene_manager.py:123-140,525-528,666-678`` -- and a default-constructed detector, which is also what the reference's benchmark does
per video, ``benchmark/__main__.py:44-61``).  The tests regenerate the frames from the seeds and require the packed / sharded flow
(``corpus.detect_corpus``: many clips in one device batch behind ``psd_score_segments_downscaled_device``) to return the same cut
lists, on the GPU (``-m gpu``: every clip) and over the CPU oracle (the smaller ones).

What this pins: the reference's Python control flow, its numpy arithmetic and its choice of downscale factor / target size (real),
on top of the restated cv2 primitives (oracle/cv2_restate.c -- PARITY UNPINNED at that boundary).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "cv2_shim"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402
from scenedetect.detectors import AdaptiveDetector, ContentDetector, HistogramDetector, ThresholdDetector  # noqa: E402
from scenedetect.scene_manager import SceneManager, compute_downscale_factor  # noqa: E402
from scenedetect.stats_manager import StatsManager  # noqa: E402

from oracle.gen_golden import MemoryStream  # noqa: E402
from pyscenedetect_amd.synth import make_clip_fast  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "corpus_default_pipeline.json")

#: name -> (seed, frames, height, width, make_clip_fast keywords); "cpu": small enough for the CPU suite
CLIPS = {
    "bbc_a": (9101, 300, 360, 640, {}, True),                          # the BBC set's resolution: factor 2.5 -> 256 x 144
    "bbc_b": (9102, 221, 360, 640, {"shot_len": (20, 60)}, True),
    "bbc_c": (9103, 64, 360, 640, {"shot_len": (17, 30)}, True),
    "qhd_a": (9104, 120, 540, 960, {}, True),                          # 3.75 -> 256 x 144
    "odd_a": (9105, 90, 270, 486, {}, True),                           # 1.898... -> 256 x 142, rows of 1458 bytes (not 16-byte multiples)
    "portrait_a": (9106, 80, 640, 360, {}, True),                      # the factor comes from the LARGER side: 2.5 -> 144 x 256
    "small_a": (9107, 70, 120, 200, {}, True),                         # narrower than 256: not resized at all
    "edge_a": (9108, 60, 300, 257, {}, True),                          # factor 1.171875 (barely above 1): 219 x 256, overlapping taps
    "noisy_a": (9113, 75, 360, 640, {"noise": 120, "shot_len": (30, 40)}, True),   # heavy noise: the downscale averages it, full-size frames do not
    "hd_a": (9109, 40, 1080, 1920, {"shot_len": (16, 18)}, False),     # 7.5 -> 256 x 144
    "hd_b": (9110, 33, 1080, 1920, {"shot_len": (16, 18)}, False),
    "uhd_a": (9111, 22, 2160, 3840, {"shot_len": (16, 18)}, False),    # 15 -> 256 x 144
    "uhd_b": (9112, 18, 2160, 3840, {"shot_len": (16, 17)}, False),
}

#: corpus name -> the reference's default-constructed detector (the AdaptiveDetector line is BASELINE configs[3]'s, spelled out)
DETECTORS = {
    "content": lambda: ContentDetector(),
    "adaptive": lambda: AdaptiveDetector(window_width=2, min_content_val=15.0),
    "hist": lambda: HistogramDetector(),
    "threshold": lambda: ThresholdDetector(),
}


def reference_default_cuts(frames, fps, make_detector):
    sm = SceneManager(None)                      # detect(): no StatsManager unless a stats file was asked for
    assert sm.auto_downscale
    sm.add_detector(make_detector())
    sm.detect_scenes(MemoryStream(frames, fps), show_progress=False)
    return [int(c.frame_num) for c in sm.get_cut_list(show_warning=False)]


def reference_default_content_val(frames, fps):
    """content_val of every frame as the default pipeline scores it (a StatsManager attached: same resize, same score)."""
    stats = StatsManager()
    sm = SceneManager(stats)
    sm.add_detector(ContentDetector())
    sm.detect_scenes(MemoryStream(frames, fps), show_progress=False)
    vals = [stats.get_metrics(i, ["content_val"])[0] if stats.metrics_exist(i, ["content_val"]) else None for i in range(len(frames))]
    return [None if v is None else float(v) for v in vals]


def main():
    out = {"fps": 25.0, "clips": {}}
    for name, (seed, n, h, w, kw, cpu) in CLIPS.items():
        frames, truth = make_clip_fast(seed, n, h, w, **kw)
        factor = compute_downscale_factor(max(w, h))
        size = [max(1, round(h / factor)), max(1, round(w / factor))] if factor > 1.0 else [h, w]
        cuts = {d: reference_default_cuts(frames, 25.0, mk) for d, mk in DETECTORS.items()}
        out["clips"][name] = {"seed": seed, "n": n, "h": h, "w": w, "kwargs": {k: list(v) if isinstance(v, tuple) else v for k, v in kw.items()},
                              "cpu": cpu, "content_val": reference_default_content_val(frames, 25.0),
                              "sum_all": int(frames.sum(dtype=np.uint64)), "factor": factor, "scored_size": size, "shot_starts": truth,
                              "cuts": cuts}
        print(name, (n, h, w), "factor", factor, "->", size, {k: len(v) for k, v in cuts.items()}, flush=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
