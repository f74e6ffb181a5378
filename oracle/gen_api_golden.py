"""Differential fixtures for the detector plug-in surface: for every constructor call below the
REFERENCE's own classes (unmodified, imported from /root/reference over the cv2 shim) report either the exception type
or the observable plug-in properties (metric keys, event_buffer_length, frames a cut may lie in the past).
tests/test_host_golden.py replays the calls against pyscenedetect_amd.

    PYTHONPATH=oracle/cv2_shim:/root/repo:/root/reference python oracle/gen_api_golden.py

TEST INFRASTRUCTURE ONLY.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "cv2_shim"), os.path.dirname(HERE), "/root/reference"]

import numpy as np  # noqa: E402

from scenedetect.common import FrameTimecode  # noqa: E402
from scenedetect.detector import FlashFilter  # noqa: E402
from scenedetect.detectors import (AdaptiveDetector, ContentDetector, HashDetector, HistogramDetector,  # noqa: E402
                                   ThresholdDetector)
from scenedetect.scene_manager import SceneManager, compute_downscale_factor  # noqa: E402

CLASSES = {"ContentDetector": ContentDetector, "AdaptiveDetector": AdaptiveDetector, "HistogramDetector": HistogramDetector,
           "ThresholdDetector": ThresholdDetector, "HashDetector": HashDetector}

CTOR_CASES = [
    ("ContentDetector", {}),
    ("ContentDetector", {"threshold": 12.5, "min_scene_len": 0}),
    ("ContentDetector", {"min_scene_len": 30}),
    ("ContentDetector", {"min_scene_len": 1.5}),
    ("ContentDetector", {"min_scene_len": "00:00:02.000"}),
    ("ContentDetector", {"min_scene_len": "48"}),
    ("ContentDetector", {"luma_only": True}),
    ("ContentDetector", {"weights": [0.5, 1.0, 2.0, 0.25]}),
    ("ContentDetector", {"kernel_size": 3}),
    ("ContentDetector", {"kernel_size": 7}),
    ("ContentDetector", {"kernel_size": 4}),
    ("ContentDetector", {"kernel_size": 1}),
    ("ContentDetector", {"kernel_size": -3}),
    ("ContentDetector", {"filter_mode": "SUPPRESS"}),
    ("ContentDetector", {"filter_mode": "SUPPRESS", "min_scene_len": 2.0}),
    ("AdaptiveDetector", {}),
    ("AdaptiveDetector", {"window_width": 1}),
    ("AdaptiveDetector", {"window_width": 5, "min_scene_len": 3}),
    ("AdaptiveDetector", {"window_width": 0}),
    ("AdaptiveDetector", {"window_width": -2}),
    ("AdaptiveDetector", {"adaptive_threshold": 1.0, "min_content_val": 0.0}),
    ("AdaptiveDetector", {"kernel_size": 6}),
    ("AdaptiveDetector", {"luma_only": True, "window_width": 3}),
    ("HistogramDetector", {}),
    ("HistogramDetector", {"bins": 16}),
    ("HistogramDetector", {"bins": 256, "threshold": 0.0}),
    ("HistogramDetector", {"threshold": 1.5}),
    ("HistogramDetector", {"threshold": -0.5}),
    ("ThresholdDetector", {}),
    ("ThresholdDetector", {"threshold": 12.7}),
    ("ThresholdDetector", {"method": "CEILING", "fade_bias": 1.0}),
    ("ThresholdDetector", {"add_final_scene": True, "min_scene_len": 0}),
    ("ThresholdDetector", {"fade_bias": -1.0}),
    ("HashDetector", {}),
    ("HashDetector", {"size": 16, "lowpass": 4}),
    ("HashDetector", {"threshold": 0.0, "min_scene_len": 1}),
]


def build(cls_name, kwargs):
    kw = dict(kwargs)
    if "weights" in kw:
        kw["weights"] = ContentDetector.Components(*kw["weights"])
    if "filter_mode" in kw:
        kw["filter_mode"] = FlashFilter.Mode[kw["filter_mode"]]
    if "method" in kw:
        kw["method"] = ThresholdDetector.Method[kw["method"]]
    return CLASSES[cls_name](**kw)


def outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001
        return {"raises": type(ex).__name__}


def describe(det):
    return {"metrics": list(det.get_metrics()), "event_buffer_length": int(det.event_buffer_length),
            "stats_manager_is_none": det.stats_manager is None}


def bad_frames():
    """process_frame on frames a detector must refuse (histogram_detector.py:80-84)."""
    tc = FrameTimecode(0, 25.0)
    cases = {"hist_uint16": ("HistogramDetector", np.zeros((8, 8, 3), np.uint16)),
             "hist_float": ("HistogramDetector", np.zeros((8, 8, 3), np.float32)),
             "hist_4ch": ("HistogramDetector", np.zeros((8, 8, 4), np.uint8))}
    out = {}
    for name, (cls, frame) in cases.items():
        out[name] = {"cls": cls, "dtype": str(frame.dtype), "shape": list(frame.shape),
                     **outcome(lambda: [c.frame_num for c in CLASSES[cls]().process_frame(tc, frame)])}
    return out


from manager_cases import manager_cases  # noqa: E402  (shared with tests/test_host_golden.py)


def main():
    out = {"ctor": [{"cls": c, "kwargs": k, **outcome(lambda: describe(build(c, k)))} for c, k in CTOR_CASES]}
    out["bad_frames"] = bad_frames()
    out["downscale_factor"] = [{"width": w, "effective": e, **outcome(lambda: compute_downscale_factor(w, e))}
                               for w in (1, 100, 255, 256, 257, 320, 512, 640, 1280, 1920, 3840, 7680)
                               for e in (256, 128, 1000)]
    # how many frames SceneManager keeps for callbacks with several detectors registered (scene_manager.py:337-352)
    def buffer_sizes():
        sm = SceneManager()
        sizes = []
        for c, k in (("ThresholdDetector", {}), ("ContentDetector", {}), ("AdaptiveDetector", {"window_width": 4}),
                     ("ContentDetector", {"min_scene_len": 40})):
            sm.add_detector(build(c, k))
            sizes.append(sm._frame_buffer_size)
        return sizes
    out["frame_buffer_sizes"] = outcome(buffer_sizes)
    # FlashFilter driven directly with random above/below-threshold sequences: which cuts come out, and WHEN
    # (merge mode emits late), for frame- and time-based lengths at several frame rates (detector.py:106-224)
    rng = np.random.default_rng(77)
    out["flash_filter"] = []
    for case in range(48):
        fps = [25.0, 29.97, 10.0, 60.0][case % 4]
        length = [15, 0, 1, 7, 0.5, 1.2, "00:00:00.300", "20", 40, 0.0, "0.8s", 3][case % 12]
        mode = "MERGE" if (case // 4) % 2 == 0 else "SUPPRESS"
        n = 220
        density = [0.02, 0.1, 0.3, 0.6][(case // 12) % 4]
        above = (rng.random(n) < density).tolist()
        def drive():
            flt = FlashFilter(FlashFilter.Mode[mode], length)
            emitted = []
            for i, a in enumerate(above):
                for c in flt.filter(FrameTimecode(i, fps), bool(a)):
                    emitted.append([i, c.frame_num])
            return {"emitted": emitted, "max_behind": int(flt.max_behind)}
        out["flash_filter"].append({"fps": fps, "length": length, "mode": mode, "above": [int(a) for a in above], **outcome(drive)})
    # the same with presentation-timestamp positions (variable frame rate: 20 / 40 / 60 / 80 ms steps at a nominal 25 fps),
    # where the filter's gaps are time differences, not frame counts
    from fractions import Fraction

    from scenedetect.common import Timecode

    out["flash_filter_pts"] = []
    for case in range(36):
        length = [15, 1, 7, 0.5, 1.2, "00:00:00.300", "20", 40, "0.8s", 3, 0.25, 25][case % 12]
        mode = "MERGE" if (case // 3) % 2 == 0 else "SUPPRESS"
        n = 200
        density = [0.05, 0.2, 0.5][case % 3]
        above = (rng.random(n) < density).tolist()
        steps = rng.choice([20, 40, 40, 40, 60, 80], size=n - 1)
        pts = [0] + [int(x) for x in np.cumsum(steps)]
        def drive_pts():
            flt = FlashFilter(FlashFilter.Mode[mode], length)
            emitted = []
            for i, a in enumerate(above):
                tc = FrameTimecode(Timecode(pts[i], Fraction(1, 1000)), 25.0)
                for c in flt.filter(tc, bool(a)):
                    emitted.append([i, c.pts])
            return {"emitted": emitted, "max_behind": int(flt.max_behind)}
        out["flash_filter_pts"].append({"length": length, "mode": mode, "above": [int(a) for a in above], "pts": pts, **outcome(drive_pts)})
    from gen_golden import CLIPS, MemoryStream  # the reference's VideoStream over frames in memory
    from pyscenedetect_amd.synth import make_clip
    from scenedetect.stats_manager import StatsManager

    seed, n, h, w, kw = CLIPS["scenes_a"]
    out["manager_ops"] = manager_cases(lambda: make_clip(seed, n, h, w, **kw)[0],
                                       lambda with_stats: SceneManager(StatsManager() if with_stats else None),
                                       lambda frames: MemoryStream(frames, 25.0), lambda: ContentDetector())
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "api_cases.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
    for c in out["ctor"]:
        print(c["cls"], c["kwargs"], c.get("raises") or c["ok"])
    print(out["bad_frames"]); print(out["frame_buffer_sizes"])


if __name__ == "__main__":
    main()
