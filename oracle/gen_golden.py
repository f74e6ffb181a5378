"""Generate tests/golden/*.json by running the UNMODIFIED reference over the cv2 shim.

Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py

For every (clip, detector configuration) the reference's own ``SceneManager.detect_scenes`` is
run on an in-memory ``VideoStream`` of seeded synthetic frames (``pyscenedetect_amd.synth``); the
resulting cut list and every per-frame metric the reference writes to its ``StatsManager`` are
stored.  The tests regenerate the same frames from the seed and require the HIP path (``-m gpu``)
and the oracle-backed host path (CPU) to reproduce them.

What this pins: the reference's Python control flow and numpy arithmetic (real), on top of the
restated cv2 primitives (oracle/cv2_restate.c -- PARITY UNPINNED at that boundary).
"""

import json
import os
import sys
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "cv2_shim"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402
import scenedetect  # noqa: E402  (the reference)
from scenedetect import FrameTimecode  # noqa: E402
from scenedetect.common import Timecode  # noqa: E402
from scenedetect.detector import FlashFilter  # noqa: E402
from scenedetect.detectors import (AdaptiveDetector, ContentDetector, HashDetector, HistogramDetector,  # noqa: E402
                                   ThresholdDetector)
from scenedetect.scene_manager import SceneManager  # noqa: E402
from scenedetect.stats_manager import StatsManager  # noqa: E402
from scenedetect.video_stream import VideoStream  # noqa: E402

from pyscenedetect_amd.synth import make_clip  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class MemoryStream(VideoStream):
    """Reference VideoStream over frames in memory."""

    BACKEND_NAME = "memory"

    def __init__(self, frames, fps=25.0):
        self._frames = frames
        self._fps = Fraction(fps).limit_denominator(10000)
        self._next = 0

    path = property(lambda self: "memory")
    name = property(lambda self: "memory")
    is_seekable = property(lambda self: True)
    frame_rate = property(lambda self: self._fps)
    duration = property(lambda self: self.base_timecode + len(self._frames))
    frame_size = property(lambda self: (self._frames.shape[2], self._frames.shape[1]))
    aspect_ratio = property(lambda self: 1.0)
    position = property(lambda self: self.base_timecode + max(0, self._next - 1))
    position_ms = property(lambda self: self.position.seconds * 1000.0)
    frame_number = property(lambda self: self._next)

    def read(self, decode=True):
        if self._next >= len(self._frames):
            return False
        f = self._frames[self._next]
        self._next += 1
        return f if decode else True

    def reset(self):
        self._next = 0

    def seek(self, target):
        self._next = int(target)


class VfrMemoryStream(MemoryStream):
    """Positions are presentation timestamps (what the reference's PyAV backend reports, backends/pyav.py): frame i
    is shown at pts[i] ticks of `time_base`; the nominal frame rate is only an average."""

    def __init__(self, frames, fps, pts, time_base):
        super().__init__(frames, fps)
        self._pts = pts
        self._tb = time_base

    position = property(lambda self: FrameTimecode(Timecode(self._pts[max(0, self._next - 1)], self._tb), self._fps))
    duration = property(lambda self: FrameTimecode(Timecode(self._pts[-1], self._tb), self._fps))


def vfr_pts(n, seed=5):
    """Irregular frame durations: 40 ms nominal, with 20 / 60 / 80 ms outliers."""
    rng = np.random.default_rng(seed)
    steps = rng.choice([20, 40, 40, 40, 40, 60, 80], size=n - 1)
    return [0] + [int(x) for x in np.cumsum(steps)]


# name -> (seed, n_frames, height, width, make_clip kwargs)
CLIPS = {
    "scenes_a": (11, 260, 72, 128, {}),
    "fades_b": (12, 220, 54, 96, {"fade_every": 2, "shot_len": (30, 50)}),
    "ragged_c": (13, 90, 37, 53, {"shot_len": (8, 20)}),   # H*W not a multiple of 16
    "wide_d": (14, 60, 180, 320, {"shot_len": (10, 25)}),  # > 256 px wide: auto-downscale kicks in
}


BIG_CLIP = (15, 56, 540, 960, {"shot_len": (16, 22)})


def uniform_clip(seed, n, h, w):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


# name -> (class name, kwargs, with_stats, auto_downscale)
CONFIGS = {
    "content_default": ("ContentDetector", {}, False),
    "content_stats": ("ContentDetector", {}, True),
    "content_edges": ("ContentDetector", {"weights": [1.0, 1.0, 1.0, 1.0], "threshold": 35.0}, True),
    "content_luma_suppress": ("ContentDetector", {"luma_only": True, "filter_mode": "SUPPRESS", "min_scene_len": 6,
                                                  "threshold": 20.0}, True),
    "content_kernel5_secs": ("ContentDetector", {"kernel_size": 5, "min_scene_len": 0.5}, True),
    "adaptive_default": ("AdaptiveDetector", {}, True),
    "adaptive_w3": ("AdaptiveDetector", {"window_width": 3, "min_content_val": 10.0, "min_scene_len": "0.4s",
                                        "adaptive_threshold": 2.5}, True),
    "hist_default": ("HistogramDetector", {}, True),
    "hist_256": ("HistogramDetector", {"bins": 256, "threshold": 0.1, "min_scene_len": 5}, True),
    "hist_100": ("HistogramDetector", {"bins": 100, "threshold": 0.3}, True),
    "hash_default": ("HashDetector", {}, True),
    "hash_16_lp2": ("HashDetector", {"size": 16, "lowpass": 2, "threshold": 0.3, "min_scene_len": 5}, True),
    "hash_8_lp4_secs": ("HashDetector", {"size": 8, "lowpass": 4, "threshold": 0.4, "min_scene_len": 0.4}, False),
    "threshold_default": ("ThresholdDetector", {}, True),
    "threshold_final": ("ThresholdDetector", {"threshold": 30, "add_final_scene": True, "fade_bias": 0.5,
                                             "min_scene_len": 4}, True),
    "threshold_ceiling": ("ThresholdDetector", {"threshold": 90, "method": "CEILING", "min_scene_len": 3,
                                               "fade_bias": -0.5}, True),
}


def build_detector(cls_name, kwargs):
    kw = dict(kwargs)
    if "weights" in kw:
        kw["weights"] = ContentDetector.Components(*kw["weights"])
    if "filter_mode" in kw:
        kw["filter_mode"] = FlashFilter.Mode[kw["filter_mode"]]
    if "method" in kw:
        kw["method"] = ThresholdDetector.Method[kw["method"]]
    return {"ContentDetector": ContentDetector, "AdaptiveDetector": AdaptiveDetector,
            "HistogramDetector": HistogramDetector, "ThresholdDetector": ThresholdDetector,
            "HashDetector": HashDetector}[cls_name](**kw)


def run(frames, cls_name, kwargs, with_stats, auto_downscale, fps=25.0, interpolation=None):
    stats = StatsManager() if with_stats else None
    sm = SceneManager(stats)
    sm.auto_downscale = auto_downscale
    if interpolation is not None:
        from scenedetect.common import Interpolation

        sm.interpolation = Interpolation[interpolation]
    det = build_detector(cls_name, kwargs)
    sm.add_detector(det)
    video = MemoryStream(frames, fps)
    n = sm.detect_scenes(video)
    cuts = [c.frame_num for c in sm.get_cut_list(show_warning=False)]
    scenes = [[a.frame_num, b.frame_num] for a, b in sm.get_scene_list()]
    metrics = {}
    if stats is not None:
        for key in det.get_metrics():
            vals = []
            for i in range(len(frames)):
                v = stats.get_metrics(i, [key])[0] if stats.metrics_exist(i, [key]) else None
                vals.append(None if v is None else float(v))
            metrics[key] = vals
    return {"frames_processed": n, "cuts": cuts, "scenes": scenes, "metrics": metrics}


def kats():
    """Known-answer traces of the pure state machines with injected scores."""
    out = {}
    tc = lambda i: FrameTimecode(i, 25.0)  # noqa: E731
    above = {10, 12, 20, 40, 41, 42, 70, 100, 103, 130}
    for mode in ("MERGE", "SUPPRESS"):
        for length in (15, 0.6, "0.6s", "15"):
            f = FlashFilter(FlashFilter.Mode[mode], length)
            emitted = []
            for i in range(160):
                emitted += [[i, c.frame_num] for c in f.filter(tc(i), i in above)]
            out[f"flash_{mode}_{length!r}"] = {"above": sorted(above), "n": 160, "emitted": emitted,
                                               "max_behind": f.max_behind}
    scores = {30: 40.0, 33: 35.0, 60: 26.9, 61: 27.0, 90: 100.0, 140: 50.0, 141: 50.0, 142: 50.0}

    class Injected(ContentDetector):
        def _calculate_frame_score(self, timecode, frame_img):
            return scores.get(timecode.frame_num, 0.0)

    d = Injected()
    emitted = []
    for i in range(200):
        emitted += [c.frame_num for c in d.process_frame(tc(i), None)]
    out["content_injected"] = {"scores": {str(k): v for k, v in scores.items()}, "n": 200, "default": 0.0, "cuts": emitted}

    ascores = {30: 40.0, 50: 16.0, 51: 16.0, 80: 14.9, 100: 30.0, 104: 30.0, 150: 200.0}

    class InjectedA(AdaptiveDetector):
        def _calculate_frame_score(self, timecode, frame_img):
            return ascores.get(timecode.frame_num, 1.0)

    d = InjectedA()
    emitted = []
    for i in range(200):
        emitted += [c.frame_num for c in d.process_frame(tc(i), None)]
    out["adaptive_injected"] = {"scores": {str(k): v for k, v in ascores.items()}, "n": 200, "default": 1.0, "cuts": emitted}
    return out


def scenarios():
    """SceneManager behaviours around the per-frame loop (seek/end_time/duration, start_in_scene,
    callbacks with look-behind, crop, frame_skip), again from the unmodified reference."""
    frames, _ = make_clip(*CLIPS["scenes_a"][:4], **CLIPS["scenes_a"][4])
    out = {}

    def manager(det, stats=False, **attrs):
        sm = SceneManager(StatsManager() if stats else None)
        sm.auto_downscale = False
        for k, v in attrs.items():
            setattr(sm, k, v)
        sm.add_detector(det)
        return sm

    def scene_nums(sm, **kw):
        return [[a.frame_num, b.frame_num] for a, b in sm.get_scene_list(**kw)]

    # window: seek + end_time
    video = MemoryStream(frames)
    video.seek(40)
    sm = manager(ContentDetector())
    n = sm.detect_scenes(video, end_time=200)
    out["window_seek40_end200"] = {"frames_processed": n, "scenes": scene_nums(sm),
                                   "cuts": [c.frame_num for c in sm.get_cut_list(show_warning=False)]}
    # duration
    video = MemoryStream(frames)
    sm = manager(ContentDetector())
    n = sm.detect_scenes(video, duration=100)
    out["duration_100"] = {"frames_processed": n, "scenes": scene_nums(sm)}
    # no cuts -> empty list unless start_in_scene
    video = MemoryStream(frames)
    sm = manager(ContentDetector())
    n = sm.detect_scenes(video, end_time=10)
    out["short_no_cuts"] = {"frames_processed": n, "scenes": scene_nums(sm), "scenes_start_in_scene": scene_nums(sm, start_in_scene=True)}
    # callbacks
    for name, det in (("content", ContentDetector()), ("adaptive", AdaptiveDetector()),
                      ("content_suppress", ContentDetector(filter_mode=FlashFilter.Mode.SUPPRESS, min_scene_len=6, threshold=20.0))):
        calls = []
        video = MemoryStream(frames)
        sm = manager(det)
        sm.detect_scenes(video, callback=lambda img, pos: calls.append([pos.frame_num, int(img.sum())]))
        out[f"callback_{name}"] = {"calls": calls, "cuts": [c.frame_num for c in sm.get_cut_list(show_warning=False)]}
    # crop (inclusive coordinates)
    video = MemoryStream(frames)
    sm = manager(ContentDetector(), stats=True, crop=(10, 5, 100, 60))
    sm.detect_scenes(video)
    cv = [sm.stats_manager.get_metrics(i, ["content_val"])[0] for i in range(len(frames))]
    out["crop_10_5_100_60"] = {"cuts": [c.frame_num for c in sm.get_cut_list(show_warning=False)],
                               "content_val": [None if v is None else float(v) for v in cv]}
    # frame_skip
    video = MemoryStream(frames)
    sm = manager(ContentDetector())
    n = sm.detect_scenes(video, frame_skip=1)
    out["frame_skip_1"] = {"frames_processed": n, "cuts": [c.frame_num for c in sm.get_cut_list(show_warning=False)],
                           "scenes": scene_nums(sm)}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    golden = {"reference_version": scenedetect.__version__, "numpy": np.__version__, "clips": {}, "configs": CONFIGS}
    for name, (seed, n, h, w, kw) in CLIPS.items():
        frames, truth = make_clip(seed, n, h, w, **kw)
        entry = {"seed": seed, "n": n, "h": h, "w": w, "kwargs": kw, "true_cuts": truth,
                 "sha_first_frame_sum": int(frames[0].sum()), "sum_all": int(frames.sum()), "results": {}}
        for cname, (cls_name, kwargs, with_stats) in CONFIGS.items():
            auto_ds = name == "wide_d"
            entry["results"][cname] = run(frames, cls_name, kwargs, with_stats, auto_ds)
            print(name, cname, entry["results"][cname]["cuts"])
        golden["clips"][name] = entry
    fr = uniform_clip(21, 40, 48, 64)
    entry = {"seed": 21, "n": 40, "h": 48, "w": 64, "uniform": True, "sum_all": int(fr.sum()), "results": {}}
    for cname in ("content_stats", "content_edges", "hist_default", "threshold_default", "adaptive_default", "hash_default"):
        cls_name, kwargs, with_stats = CONFIGS[cname]
        entry["results"][cname] = run(fr, cls_name, kwargs, with_stats, False)
    golden["clips"]["uniform_u"] = entry
    # the non-default downscale filters the device implements (scene_manager.py:265-272, common.py:148-160)
    frames, _ = make_clip(*CLIPS["wide_d"][:4], **CLIPS["wide_d"][4])
    golden["interp"] = {}
    for mode in ("NEAREST", "AREA", "LANCZOS4", "CUBIC"):     # (CUBIC: the shim's default form, "sse")
        golden["interp"][mode] = {}
        for cname in ("content_stats", "hist_default", "hash_default"):
            cls_name, kwargs, with_stats = CONFIGS[cname]
            golden["interp"][mode][cname] = run(frames, cls_name, kwargs, with_stats, True, interpolation=mode)
            print("wide_d", mode, cname, golden["interp"][mode][cname]["cuts"])
    # a frame size where the default downscale skips most source rows (960 x 540 -> 256 x 144, factor 3.75): the host feeder
    # of the GPU path uploads only the rows that carry taps there (psd_upload_rows) -- the reference sees whole frames
    seed, n, h, w, kw = BIG_CLIP
    frames, truth = make_clip(seed, n, h, w, **kw)
    golden["downscale_rows"] = {"clip": {"seed": seed, "n": n, "h": h, "w": w, "kwargs": kw, "true_cuts": truth,
                                         "sum_all": int(frames.sum())}, "results": {}}
    for mode in ("LINEAR", "NEAREST", "LANCZOS4", "CUBIC"):
        golden["downscale_rows"]["results"][mode] = {}
        for cname in ("content_stats", "content_edges", "adaptive_default", "hist_default", "hash_default", "threshold_default"):
            cls_name, kwargs, with_stats = CONFIGS[cname]
            golden["downscale_rows"]["results"][mode][cname] = run(frames, cls_name, kwargs, with_stats, True, interpolation=mode)
            print("big_e", mode, cname, golden["downscale_rows"]["results"][mode][cname]["cuts"])
    del frames
    # variable frame rate: PTS-backed positions through the unmodified SceneManager / detectors
    frames, _ = make_clip(*CLIPS["scenes_a"][:4], **CLIPS["scenes_a"][4])
    pts = vfr_pts(len(frames))
    golden["vfr"] = {"clip": "scenes_a", "pts_seed": 5, "time_base": [1, 1000], "fps": 25.0, "results": {}}
    for cname in ("content_default", "content_kernel5_secs", "adaptive_w3", "hist_default", "threshold_final", "hash_8_lp4_secs"):
        cls_name, kwargs, with_stats = CONFIGS[cname]
        sm = SceneManager(StatsManager() if with_stats else None)
        sm.auto_downscale = False
        sm.add_detector(build_detector(cls_name, kwargs))
        n = sm.detect_scenes(VfrMemoryStream(frames, 25.0, pts, Fraction(1, 1000)))
        cuts = sm.get_cut_list(show_warning=False)
        golden["vfr"]["results"][cname] = {
            "frames_processed": n,
            "cuts": [[c.frame_num, c.pts, c.seconds, c.get_timecode()] for c in cuts],
            "scenes": [[a.pts, b.pts, a.get_timecode(), b.get_timecode()] for a, b in sm.get_scene_list()]}
        print("vfr", cname, [c.pts for c in cuts])
    golden["kats"] = kats()
    golden["scenarios"] = scenarios()
    path = os.path.join(OUT, "reference_runs.json")
    with open(path, "w") as f:
        json.dump(golden, f, indent=None, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
