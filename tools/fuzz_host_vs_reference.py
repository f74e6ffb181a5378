"""Differential fuzz of the HOST side: the unmodified reference (``/root/reference/scenedetect`` over the cv2 shim) against the
mirror (``pyscenedetect_amd`` over the CPU oracle engine), on random clips, detector sets, detector parameters and
SceneManager settings.  Build container only (the reference is not on the GPU box); no GPU involved.

    python tools/fuzz_host_vs_reference.py [--seconds 120] [--seed 1] [--wide] [--tiny] [--force ...] [--verbose]
                                           [--cross] [--plug] [--guest] [--sim] [--binding]      |      --engines (GPU box)

INGREDIENTS.  Every case draws a small synthetic clip (``pyscenedetect_amd.synth.make_clip``, sometimes with fades to black, sometimes
uniform noise; ``--tiny``: frames of a few pixels, hundreds of frames), a frame rate, one to three detectors with random constructor
arguments (thresholds, ``min_scene_len`` as frames / seconds / a timecode string -- ``--wide``: as a FrameTimecode / Timecode object --,
weights with and without the edge term -- ``--wide``: fractional, negative, all zero --, ``luma_only``, ``kernel_size``, filter modes,
window widths, bins, fade bias, FLOOR / CEILING, ``add_final_scene``, hash sizes), a StatsManager or none, and SceneManager settings
(auto / manual downscale with each interpolation the mirror supports, crop, ``frame_skip``, ``end_time`` / ``duration`` as frames or
seconds, ``start_in_scene``, presentation timestamps, a ``callback``, a few frames of another size, exceptions from the caller's
callback or the stream's ``read``, inputs both sides must refuse).  MODES: one manager run; the plug-in API by hand (``process_frame`` /
``post_process``); one manager on two videos with or without ``clear()``; a second detection from the saved stats CSV; with ``--wide``
also detection from a seek position and in pieces (a detector joining in between), the package's ``detect()``, frame layouts
(``rgb[..., ::-1]`` views, windows of padded buffers, read-only arrays), a detector of the caller's own on the plug-in API, and the SAME
detector objects under a manager for half of the clip and fed by hand -- or under another manager -- for the other half.

SIDES.  The plain reference is always side one, the mirror over the oracle engine side two.  ``--cross``: the mirror's manager and
detectors reading one of the REFERENCE's VideoStream objects.  ``--plug``: the mirror's detectors registered with the REFERENCE's own
SceneManager / StatsManager.  ``--guest``: a detector derived from the reference's SceneDetector ABC under the mirror's manager.
``--sim``: the mirror over a host-memory stand-in of the DEVICE engine (feeder, tap rows, slots, halo frame, resident per-frame
buffers): the Python half of the GPU path.  ``--binding``: the reference with INTEGRATION.md B's seams bound to the C-ABI (its CPU
build).  ``--engines`` (GPU box, no reference needed): the mirror over the HIP engine against the mirror over the oracle engine.

OUTCOME, required to be the SAME on every side: frames processed, cut list, scene list, every per-frame metric bit for bit AND the type
of its values, the frames handed to the callback and to a plug-in detector (number, shape, checksum), the text of the saved stats CSV,
the second detection from it, the stream's position afterwards, the detectors' public attributes afterwards, every warning (category
and text) and every log record (level and text) -- or the same exception type with the same text.  Prints one JSON line: cases, by
detector, the first mismatches with the seed and case number that reproduce them."""
import argparse
import json
import os
import sys
import time
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle", "cv2_shim"), ROOT, "/root/reference"]

import numpy as np  # noqa: E402

try:        # the reference side: the build container only (``--engines`` on the GPU box compares two engines behind the mirror instead)
    import scenedetect as ref  # noqa: E402  (the reference, unmodified)
    from scenedetect.common import Interpolation as RefInterpolation  # noqa: E402
    from scenedetect.detector import FlashFilter as RefFlashFilter  # noqa: E402
    from scenedetect.detectors import (AdaptiveDetector as RefAdaptive, ContentDetector as RefContent, HashDetector as RefHash,  # noqa: E402
                                       HistogramDetector as RefHistogram, ThresholdDetector as RefThreshold)
    from scenedetect.scene_manager import SceneManager as RefSceneManager  # noqa: E402
    from scenedetect.stats_manager import StatsManager as RefStatsManager  # noqa: E402
    from oracle.gen_golden import MemoryStream, VfrMemoryStream  # noqa: E402

    REF_CLASSES = {"ContentDetector": RefContent, "AdaptiveDetector": RefAdaptive, "HistogramDetector": RefHistogram,
                   "ThresholdDetector": RefThreshold, "HashDetector": RefHash}
except ImportError:
    ref = None

import pyscenedetect_amd as psd  # noqa: E402
from oracle.detectors_np import OracleEngine  # noqa: E402
from pyscenedetect_amd.synth import make_clip  # noqa: E402


def draw_min_scene_len(rng):
    if WIDE and rng.integers(0, 6) == 0:      # a TimecodeLike that is an object: FrameTimecode (frames or seconds at its own rate) or Timecode
        k = int(rng.integers(0, 3))
        rate = [25.0, 30.0, 23.976][int(rng.integers(0, 3))]
        if k == 0:
            return ("FrameTimecode", int(rng.integers(0, 30)), rate)
        if k == 1:
            return ("FrameTimecode", float(round(rng.uniform(0.05, 1.5), 3)), rate)
        return ("Timecode", int(rng.integers(0, 1500)), 1000)
    k = int(rng.integers(0, 5))
    if k == 0:
        return int(rng.integers(0, 30))
    if k == 1:
        return float(round(rng.uniform(0.05, 1.5), 3))
    if k == 2:
        return "%.3fs" % rng.uniform(0.05, 1.2)
    if k == 3:
        return "00:00:%06.3f" % rng.uniform(0.04, 1.5)
    return 15


def draw_detector(rng):
    name = ["ContentDetector", "AdaptiveDetector", "HistogramDetector", "ThresholdDetector", "HashDetector"][int(rng.integers(0, 5))]
    kw = {}
    if rng.integers(0, 4):
        kw["min_scene_len"] = draw_min_scene_len(rng)
    if name in ("ContentDetector", "AdaptiveDetector"):
        if rng.integers(0, 2):
            w = [float(rng.integers(0, 3)) for _ in range(3)] + [float(rng.integers(0, 2))]
            if WIDE and rng.integers(0, 2):      # fractional and negative weights: the score divides by the sum of their magnitudes
                w = [float(round(rng.uniform(-1.5, 2.5), 2)) if rng.integers(0, 4) else 0.0 for _ in range(4)]
            if sum(abs(x) for x in w) == 0 and not (WIDE and rng.integers(0, 2)):      # (all zero: NaN scores in the reference, no exception)
                w[2] = 1.0
            kw["weights"] = w
        if rng.integers(0, 4) == 0:
            kw["luma_only"] = True
        if rng.integers(0, 3) == 0:
            kw["kernel_size"] = int(rng.choice([3, 5, 7, 9, 11, 15, 21, 31] if WIDE else [3, 5, 7, 9]))
    if name == "ContentDetector":
        if rng.integers(0, 3):
            kw["threshold"] = float(round(rng.uniform(4.0, 70.0), 2))
        if rng.integers(0, 3) == 0:
            kw["filter_mode"] = ["MERGE", "SUPPRESS"][int(rng.integers(0, 2))]
    elif name == "AdaptiveDetector":
        if rng.integers(0, 2):
            kw["adaptive_threshold"] = float(round(rng.uniform(1.2, 6.0), 2))
        if rng.integers(0, 2):
            kw["window_width"] = int(rng.integers(1, 5))
        if rng.integers(0, 2):
            kw["min_content_val"] = float(round(rng.uniform(2.0, 30.0), 2))
    elif name == "HistogramDetector":
        if rng.integers(0, 2):
            kw["threshold"] = float(round(rng.uniform(0.01, 0.6), 3))
        if rng.integers(0, 2):
            kw["bins"] = int(rng.choice([16, 32, 64, 100, 128, 200, 256]))
    elif name == "ThresholdDetector":
        if rng.integers(0, 2):
            kw["threshold"] = float(rng.integers(3, 120)) if rng.integers(0, 2) else int(rng.integers(3, 120))
            if WIDE and rng.integers(0, 3) == 0:     # a threshold with a fraction (the reference truncates it to an int)
                kw["threshold"] = float(round(rng.uniform(0.0, 140.0), 2))
        if rng.integers(0, 2):
            kw["fade_bias"] = float(round(rng.uniform(-1.0, 1.0), 2))
        if rng.integers(0, 2):
            kw["add_final_scene"] = True
        if rng.integers(0, 3) == 0:
            kw["method"] = ["FLOOR", "CEILING"][int(rng.integers(0, 2))]
    else:
        if rng.integers(0, 2):
            kw["threshold"] = float(round(rng.uniform(0.1, 0.6), 3))
        if rng.integers(0, 2):
            kw["size"] = int(rng.choice([2, 3, 4, 8, 12, 16, 32] if WIDE else [8, 16]))
        if rng.integers(0, 2):
            kw["lowpass"] = int(rng.choice([1, 2, 3, 4, 5] if WIDE else [1, 2, 4]))
    return name, kw


_PLUGINS = {}


def plugin_class(base):
    """A detector of the caller's own, written against the plug-in API only (detector.py:37-103): it sees frames, keeps a metric,
    cuts on a jump of the frame's mean and looks ``behind`` frames back."""
    if base not in _PLUGINS:
        import zlib

        class MeanJump(base):
            def __init__(self, jump=12.0, behind=0):
                super().__init__()
                self.jump, self.behind, self.last, self.seen = jump, behind, None, []

            def process_frame(self, timecode, frame_img):
                self.seen.append([int(timecode.frame_num), list(frame_img.shape), zlib.crc32(np.ascontiguousarray(frame_img).tobytes())])
                mean = float(np.asarray(frame_img, dtype=np.float64).mean())
                if self.stats_manager is not None:
                    self.stats_manager.set_metrics(timecode, {"mean_jump": mean})
                cut = self.last is not None and abs(mean - self.last[1]) >= self.jump
                out = [self.last[0]] if cut and self.behind else [timecode] if cut else []
                self.last = (timecode, mean)
                return out

            def get_metrics(self):
                return ["mean_jump"]

            @property
            def event_buffer_length(self):
                return self.behind

        _PLUGINS[base] = MeanJump
    return _PLUGINS[base]


def build(side, name, kw, engine):
    kw = dict(kw)
    if isinstance(kw.get("min_scene_len"), tuple):      # the object form, in the class of the side that builds the detector
        kind, value, scale = kw["min_scene_len"]
        # (the mixed set-ups mix here too: whoever keeps part of the reference builds such objects with ITS classes)
        mod = psd if side == "mirror" else ref.common
        kw["min_scene_len"] = (mod.FrameTimecode(value, scale) if kind == "FrameTimecode" else mod.Timecode(value, Fraction(1, scale)))
    if name == "MeanJump":
        return plugin_class(ref.detector.SceneDetector if side in ("ref", "plug", "guest", "guest_cross") else psd.SceneDetector)(**kw)
    if side == "ref":
        cls = REF_CLASSES[name]
        if "weights" in kw:
            kw["weights"] = RefContent.Components(*kw["weights"])
        if "filter_mode" in kw:
            kw["filter_mode"] = RefFlashFilter.Mode[kw["filter_mode"]]
        if "method" in kw:
            kw["method"] = RefThreshold.Method[kw["method"]]
        return cls(**kw)
    cls = getattr(psd, name)
    if "weights" in kw:
        kw["weights"] = psd.ContentDetector.Components(*kw["weights"])
    if "filter_mode" in kw:
        kw["filter_mode"] = psd.FlashFilter.Mode[kw["filter_mode"]]
    if "method" in kw:
        kw["method"] = psd.ThresholdDetector.Method[kw["method"]]
    return cls(engine=engine, **kw)


WIDE = False     # --wide: detector arguments from wider ranges (fractional / negative weights, larger kernels, odd hash sizes, ...)
TINY = False     # --tiny: frames of a few pixels, clips of hundreds of frames: the decision logic (filters, windows, fades) per second


def draw_clip(rng):
    if TINY:
        h, w, n = int(rng.integers(1, 7)), int(rng.integers(1, 9)), int(rng.integers(40, 700))
        kind = int(rng.integers(0, 4))
        lvl = rng.integers(0, 256, (n, 1, 1, 3))
        if kind == 0:      # shots: a level per shot + noise
            cuts = np.sort(rng.integers(0, n, int(rng.integers(1, 40))))
            lvl = rng.integers(0, 256, (len(cuts) + 1, 1, 1, 3))[np.searchsorted(cuts, np.arange(n), side="right")]
        elif kind == 1:    # fades in and out of black
            t = np.arange(n)
            lvl = (np.abs(np.sin(t / rng.uniform(5, 60))) * rng.integers(20, 256)).astype(int)[:, None, None, None] * np.ones(3, int)
        elif kind == 2:    # slow drift with flashes
            lvl = np.cumsum(rng.integers(-3, 4, (n, 1, 1, 3)), axis=0) + 128
            flashes = rng.random(n) < 0.03
            lvl[flashes] = 255
        noise = rng.integers(-int(rng.integers(0, 6)), int(rng.integers(1, 6)), (n, h, w, 3))
        return np.clip(lvl + noise, 0, 255).astype(np.uint8)
    h = int(rng.choice([24, 36, 37, 48, 54, 72]))
    w = int(rng.choice([32, 53, 64, 80, 96, 128, 160, 300, 520, 640]))
    n = int(rng.integers(12, 110))
    if WIDE and rng.integers(0, 25) == 0:      # now and then frames large enough for the feeder's tap-row uploads (factor 3.75 .. 7.5)
        h, w, n = int(rng.choice([135, 180, 270])), int(rng.choice([960, 1280, 1920])), int(rng.integers(12, 24))
    kind = int(rng.integers(0, 5))
    seed = int(rng.integers(0, 1 << 30))
    if kind == 0:
        frames = np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    else:
        lo = int(rng.integers(3, 12))
        kw = {"shot_len": (lo, lo + int(rng.integers(2, 25)))}
        if kind == 1:
            kw["fade_every"] = int(rng.integers(1, 4))
        frames, _ = make_clip(seed, n, h, w, **kw)
        if kind == 2:      # a dark stretch for the fade detector
            a = int(rng.integers(0, n))
            frames[a:a + int(rng.integers(2, 20))] //= int(rng.choice([8, 32, 255]))
    return frames


def draw_case(rng):
    frames = draw_clip(rng)
    n, h, w = frames.shape[:3]
    fps = [25.0, 30.0, 24.0, 29.97, 23.976, 60.0, 12.5][int(rng.integers(0, 7))]
    dets = [draw_detector(rng) for _ in range(int(rng.choice([1, 1, 1, 2, 3])))]
    if WIDE and rng.integers(0, 5) == 0:      # a detector of the caller's own: beside the others, or alone
        own = ("MeanJump", {"jump": float(round(rng.uniform(2.0, 40.0), 1)), "behind": int(rng.choice([0, 0, 1, 4]))})
        dets = [own] if rng.integers(0, 2) else dets + [own]
    sm = {"stats": bool(rng.integers(0, 3)), "auto_downscale": bool(rng.integers(0, 2))}
    if not sm["auto_downscale"] and rng.integers(0, 3) == 0:
        sm["downscale"] = int(rng.integers(1, 4))
    if rng.integers(0, 3) == 0:
        sm["interpolation"] = ["LINEAR", "NEAREST", "AREA", "LANCZOS4"][int(rng.integers(0, 4))]
        if sm["interpolation"] == "LANCZOS4" and (w + h) % 2:      # (CUBIC, accepted since round 6, without another draw: earlier seeds keep their cases)
            sm["interpolation"] = "CUBIC"
    if rng.integers(0, 5) == 0 and w > 40 and h > 30:
        x0, y0 = int(rng.integers(0, w // 3)), int(rng.integers(0, h // 3))
        sm["crop"] = (x0, y0, int(rng.integers(x0 + 17, w)), int(rng.integers(y0 + 17, h)))
    if rng.integers(0, 5) == 0:
        sm["frame_skip"] = int(rng.integers(1, 3))
        sm["stats"] = False           # (the reference refuses frame_skip with a StatsManager)
    k = int(rng.integers(0, 8))
    if k == 0:
        sm["end_time"] = int(rng.integers(1, n + 10))
    elif k == 1:
        sm["end_time"] = float(round(rng.uniform(0.1, (n + 5) / fps), 3))
    elif k == 2:
        sm["duration"] = int(rng.integers(1, n + 10))
    elif k == 3:
        sm["duration"] = float(round(rng.uniform(0.1, (n + 5) / fps), 3))
    sm["start_in_scene"] = bool(rng.integers(0, 2))
    if rng.integers(0, 7) == 0:      # presentation timestamps instead of a constant frame rate (the reference's PyAV backend)
        steps = rng.choice([20, 40, 40, 40, 40, 60, 80], size=n - 1)
        sm["pts"] = [0] + [int(x) for x in np.cumsum(steps)]
    k = int(rng.integers(0, 10))
    if k == 0:
        sm["mode"] = "per_frame"         # the plug-in API itself: process_frame() per frame and post_process(), no SceneManager
    elif k == 1:
        sm["mode"] = "reuse"             # one manager, two videos: with clear() in between or without
        sm["clear_between"] = bool(rng.integers(0, 2))
        if WIDE and rng.integers(0, 4) == 0:
            sm["other_size2"] = True
        if WIDE and rng.integers(0, 2):
            sm["pts2"] = [0] + [int(x) for x in np.cumsum(rng.choice([20, 40, 40, 40, 40, 60, 80], size=n - 1))]
    if rng.integers(0, 4) == 0:
        sm["callback"] = True            # detect_scenes(callback=...): which frames it is handed, when, with what picture
    if sm["stats"] and rng.integers(0, 3) == 0:
        sm["second_pass"] = True         # the metrics saved to CSV, loaded into a fresh StatsManager, detection again from the cache
        if rng.integers(0, 2):           # ... by ANOTHER set of detectors: some metrics cached, some not
            sm["second_dets"] = [draw_detector(rng) for _ in range(int(rng.choice([1, 2, 3])))]
    if rng.integers(0, 12) == 0:         # a few frames of another size in the stream (the reference logs an error and skips them)
        k = int(rng.integers(1, 4))
        sm["odd_frames"] = sorted({int(x) for x in rng.integers(1, n, k)})
    if rng.integers(0, 25) == 0:
        sm["fail_at"] = ("callback" if rng.integers(0, 2) else "read", int(rng.integers(0, n)))   # an exception from the caller's side
        if sm["fail_at"][0] == "callback":
            sm["callback"] = True
    if WIDE and rng.integers(0, 5) == 0:
        sm["layout"] = ["negstride", "padded", "readonly"][int(rng.integers(0, 3))]
    if WIDE and rng.integers(0, 4) == 0:
        sm["seek"] = int(rng.integers(0, n + 1))              # detection starts somewhere inside the video (video.seek before detect_scenes)
    if WIDE and rng.integers(0, 4) == 0:                  # detection in pieces: detect_scenes(duration=...) calls in a row on one video
        sm["chunks"] = [int(rng.integers(1, max(2, n // 2))) for _ in range(int(rng.integers(1, 4)))]
    if WIDE and rng.integers(0, 14) == 0:      # one set of detectors under a manager for half of the clip and fed by hand for the other
        sm["mode"] = "mixed"
        sm["manager_first"] = bool(rng.integers(0, 2))
        if rng.integers(0, 3) == 0:
            sm["two_managers"] = True
        if rng.integers(0, 2):      # round 6: behind the manager's downscale too -- by hand the detectors then get DOWNSCALED frames,
            sm["mixed_downscale"] = True      # which is what the reference's detectors hold after a manager's pass (its rule, both sides)
        sm.pop("pts", None)
    if WIDE and "chunks" in sm and rng.integers(0, 3) == 0:
        sm["add_between"] = draw_detector(rng)            # a detector that joins after the first piece
        if rng.integers(0, 3) == 0:
            sm["drop_between"] = True                     # ... after clear_detectors()
    if WIDE and rng.integers(0, 12) == 0:                 # the package's detect() (scenedetect/__init__.py:110-219) on an opened video
        sm["mode"] = "detect"
        k = int(rng.integers(0, 4))
        if k:
            sm["start_time"] = [None, int(rng.integers(0, n)), float(round(rng.uniform(0, n / fps), 3)),
                                "00:00:%06.3f" % rng.uniform(0, n / fps)][k]
        sm.pop("end_time", None)
        k = int(rng.integers(0, 5))
        if k:
            sm["end_time"] = [None, int(rng.integers(0, n + 5)), float(round(rng.uniform(0, (n + 5) / fps), 3)),
                              "00:00:%06.3f" % rng.uniform(0, (n + 5) / fps), "%.2fs" % rng.uniform(0, (n + 5) / fps)][k]
    if rng.integers(0, 12) == 0:     # things both sides must refuse (or accept) alike
        k = int(rng.integers(0, 8))
        if k == 0:
            sm["crop"] = (w + 3, 0, w + 20, h - 1)                 # starts outside the frame
        elif k == 1:
            sm["end_time"], sm["duration"] = 10, 10                # both
        elif k == 2:
            sm["end_time"] = -1
        elif k == 3:
            sm["frame_skip"], sm["stats"] = 1, True                # the reference refuses this combination
        elif k == 4:
            dets[0] = ("ContentDetector", {"kernel_size": int(rng.choice([2, 4, 1, 0]))})
        elif k == 5:
            dets[0] = ("AdaptiveDetector", {"window_width": int(rng.choice([0, -1]))})
        elif k == 6:
            sm["crop"] = (5, 5, 2, 2) if w > 8 and h > 8 else (0, 0, 0, 0)          # corners swapped
        else:
            sm["crop"] = (0, 0, w - 1, h - 1)                      # ends exactly at the border
    return frames, fps, dets, sm


class _Frames:
    """The clip as the streams index it, with a few frames of another size and, maybe, a read that fails."""

    def __init__(self, frames, odd, fail_read):
        self.frames, self.odd, self.fail_read = frames, set(odd), fail_read
        self.shape = frames.shape

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return _Frames(self.frames[i], [], None)
        if self.fail_read is not None and i == self.fail_read:
            raise OSError("decoder failure (fuzz)")
        f = self.frames[i]
        return np.ascontiguousarray(f[: max(1, f.shape[0] - 3), : max(1, f.shape[1] - 5)]) if i in self.odd else f


def run_side(side, frames, fps, dets, cfg, engine):
    if cfg.get("layout") == "negstride":      # what `rgb[..., ::-1]` hands a detector: the same pixels, the channel axis walked backwards
        frames = np.ascontiguousarray(frames[..., ::-1])[..., ::-1]
    elif cfg.get("layout") == "padded":       # frames that are windows of larger buffers (a decoder's padded planes)
        big = np.zeros((frames.shape[0], frames.shape[1] + 2, frames.shape[2] + 3, 3), np.uint8)
        big[:, 1:-1, 2:-1] = frames
        frames = big[:, 1:-1, 2:-1]
    elif cfg.get("layout") == "readonly":
        frames = frames.copy()
        frames.setflags(write=False)
    if cfg.get("odd_frames") or (cfg.get("fail_at") or ("", 0))[0] == "read":
        frames = _Frames(frames, cfg.get("odd_frames", []), cfg["fail_at"][1] if (cfg.get("fail_at") or ("", 0))[0] == "read" else None)
    # (one frame rate for both sides: the reference-side stream class of oracle/gen_golden.py turns a float into
    #  Fraction(fps).limit_denominator(10000), FrameTimecode snaps 23.976 to 24000/1001 -- both right, not the same clock)
    fps = Fraction(fps).limit_denominator(10000)
    if side in ("ref", "plug"):      # "plug": the mirror's DETECTORS under the reference's own SceneManager, StatsManager and stream
        stats = RefStatsManager() if cfg["stats"] else None
        sm = RefSceneManager(stats)
        video = VfrMemoryStream(frames, fps, cfg["pts"], Fraction(1, 1000)) if "pts" in cfg else MemoryStream(frames, fps)
        interp = RefInterpolation
    else:
        stats = psd.StatsManager() if cfg["stats"] else None
        sm = psd.SceneManager(stats, engine=engine, batch_frames=int(cfg.get("batch_frames", 64)))
        video = (psd.ArrayVideoStream(frames, fps, pts=cfg["pts"], time_base=Fraction(1, 1000)) if "pts" in cfg
                 else psd.ArrayVideoStream(frames, fps))
        if side in ("cross", "guest_cross"):      # the mirror's manager and detectors over one of the REFERENCE's streams (its timecodes, its positions)
            video = VfrMemoryStream(frames, fps, cfg["pts"], Fraction(1, 1000)) if "pts" in cfg else MemoryStream(frames, fps)
        interp = psd.Interpolation
    def configure(sm):
        sm.auto_downscale = cfg["auto_downscale"]
        if "downscale" in cfg:
            sm.downscale = cfg["downscale"]
        if "interpolation" in cfg:
            sm.interpolation = interp[cfg["interpolation"]]
        if "crop" in cfg:
            sm.crop = cfg["crop"]
        built = [build(side, name, kw, engine) for name, kw in dets]
        for d in built:
            sm.add_detector(d)
        return built

    def detect(sm, video):
        kwargs = {k: cfg[k] for k in ("frame_skip", "end_time", "duration") if k in cfg}
        seen = []
        if cfg.get("callback"):
            import zlib

            fail_cb = cfg["fail_at"][1] if (cfg.get("fail_at") or ("", 0))[0] == "callback" else None

            def cb(img, pos):
                seen.append([int(pos), list(img.shape), zlib.crc32(np.ascontiguousarray(img).tobytes())])
                if fail_cb is not None and len(seen) > fail_cb % 5:
                    raise KeyError("callback failure (fuzz)")

            kwargs["callback"] = cb
        n = sm.detect_scenes(video, **kwargs)
        cuts = [c.frame_num for c in sm.get_cut_list(show_warning=False)]
        scenes = [[a.frame_num, b.frame_num] for a, b in sm.get_scene_list(start_in_scene=cfg["start_in_scene"])]
        return n, cuts, scenes, seen

    def metrics_of(stats, built):
        metrics = {}
        if stats is not None:
            for d in built:
                for key in d.get_metrics():
                    vals = []
                    for i in range(len(frames)):
                        v = stats.get_metrics(i, [key])[0] if stats.metrics_exist(i, [key]) else None
                        vals.append(None if v is None else float(v))
                    metrics[key] = vals
                    metrics[key + " (types)"] = sorted({type(stats.get_metrics(i, [key])[0]).__name__ for i in range(len(frames))
                                                        if stats.metrics_exist(i, [key])})
        return metrics

    if cfg.get("mode") == "per_frame":
        # SceneDetector.process_frame(timecode, frame) / post_process(timecode) as a caller of the plug-in API uses them
        # (detector.py:48-73); a StatsManager is attached the way add_detector attaches it (scene_manager.py:337-352)
        built = [build(side, name, kw, engine) for name, kw in dets]
        for d in built:
            if stats is not None:
                d.stats_manager = stats
                stats.register_metrics(d.get_metrics())
        emitted = []
        tc_cls = ref.FrameTimecode if side in ("ref", "plug") else psd.FrameTimecode
        base = video.base_timecode
        for i in range(len(frames)):
            tc = (base + i) if "pts" not in cfg else tc_cls(cfg["pts"][i] / 1000.0, fps)
            for j, d in enumerate(built):
                got = d.process_frame(tc, frames[i])
                if got:
                    emitted.append([i, j, [c.frame_num for c in got]])
        last = (base + (len(frames) - 1)) if "pts" not in cfg else tc_cls(cfg["pts"][-1] / 1000.0, fps)
        for j, d in enumerate(built):
            got = d.post_process(last)
            if got:
                emitted.append(["post", j, [c.frame_num for c in got]])
        return {"frames_processed": len(frames), "cuts": emitted, "scenes": [], "metrics": metrics_of(stats, built), "callback": [],
                "event_buffer": [int(d.event_buffer_length) for d in built]}
    if cfg.get("mode") == "detect":
        import tempfile

        det = build(side, dets[0][0], dets[0][1], engine)
        kwargs = {k: cfg[k] for k in ("start_time", "end_time") if k in cfg}
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "stats.csv") if cfg["stats"] else None
            if side in ("ref", "plug"):
                opened, ref.open_video = ref.open_video, (lambda *a, **k: video)
                try:
                    scenes = ref.detect("memory", det, stats_file_path=path, start_in_scene=cfg["start_in_scene"], **kwargs)
                finally:
                    ref.open_video = opened
            else:
                scenes = psd.detect(video, det, stats_file_path=path, start_in_scene=cfg["start_in_scene"], engine=engine, **kwargs)
            text = ""
            if path:
                with open(path) as f:
                    text = f.read()
            return {"scenes": [[a.frame_num, b.frame_num] for a, b in scenes], "csv": text,
                    "stream": [video.frame_number, video.position.frame_num]}
    if cfg.get("mode") == "mixed":
        # (both halves must see frames of one size: no crop; no aborted runs; a downscale only with `mixed_downscale`, where the by-hand
        #  half is fed the frames the manager's half scored -- resized by the manager's own factor and interpolation)
        keep_ds = bool(cfg.get("mixed_downscale"))
        cfg = {k: v for k, v in cfg.items() if k not in ("crop", "frame_skip", "fail_at", "odd_frames", "end_time", "duration") and (keep_ds or k != "downscale")}
        if not keep_ds:
            cfg["auto_downscale"] = False
        # SceneManager.detect_scenes on one half of the clip, SceneDetector.process_frame by hand on the other, the SAME detector objects:
        # what a detector derived from the last frame it saw goes with it (reference content_detector.py:189)
        built = configure(sm)
        half = len(frames) // 2
        tc_cls = ref.FrameTimecode if side in ("ref", "plug") else psd.FrameTimecode
        emitted = []

        hh, ww = frames[0].shape[:2]
        ds_factor = (psd.compute_downscale_factor(max(ww, hh)) if sm.auto_downscale else sm.downscale)
        ds_interp = int(sm.interpolation.value)

        def as_scored(f):
            if not ds_factor > 1.0:
                return f
            import cv2  # the oracle's shim

            return cv2.resize(f, (max(1, round(ww / ds_factor)), max(1, round(hh / ds_factor))), interpolation=ds_interp)

        def by_hand(lo, hi):
            for i in range(lo, hi):
                for j, d in enumerate(built):
                    got = d.process_frame(tc_cls(i, fps), as_scored(frames[i]))
                    if got:
                        emitted.append([i, j, [c.frame_num for c in got]])

        if cfg.get("two_managers"):      # ... or under ANOTHER manager for the other half (a fresh one, the same detector objects)
            n = sm.detect_scenes(video, duration=half)
            other = (RefSceneManager(stats) if side in ("ref", "plug")
                     else psd.SceneManager(stats, engine=engine, batch_frames=int(cfg.get("batch_frames", 64))))
            other.auto_downscale = sm.auto_downscale
            if not sm.auto_downscale:
                other.downscale = sm.downscale
            other.interpolation = sm.interpolation
            for d in built:
                other.add_detector(d)
            n += other.detect_scenes(video)
            emitted = [c.frame_num for c in other.get_cut_list(show_warning=False)]
        elif cfg["manager_first"]:
            n = sm.detect_scenes(video, duration=half)
            by_hand(half, len(frames))
        else:
            by_hand(0, half)
            video.seek(half)
            n = sm.detect_scenes(video)
        return {"frames_processed": n, "cuts": [c.frame_num for c in sm.get_cut_list(show_warning=False)], "by_hand": emitted,
                "metrics": metrics_of(stats, built), "public_state": [public_state(d) for d in built]}
    built = configure(sm)
    pieces = []
    if "seek" in cfg:
        video.seek(cfg["seek"])
    for d in cfg.get("chunks", []):
        got = sm.detect_scenes(video, duration=d, frame_skip=cfg.get("frame_skip", 0))
        pieces.append([got, [c.frame_num for c in sm.get_cut_list(show_warning=False)],
                       [[a.frame_num, b.frame_num] for a, b in sm.get_scene_list(start_in_scene=cfg["start_in_scene"])],
                       video.frame_number, video.position.frame_num])
        if "add_between" in cfg and len(pieces) == 1:
            if cfg.get("drop_between"):
                sm.clear_detectors()
                built = []
            built.append(build(side, cfg["add_between"][0], cfg["add_between"][1], engine))
            sm.add_detector(built[-1])
    n, cuts, scenes, seen = detect(sm, video)
    out = {"frames_processed": n, "cuts": cuts, "scenes": scenes, "metrics": metrics_of(stats, built), "callback": seen,
           "pieces": pieces, "stream": [video.frame_number, video.position.frame_num],
           "plugin_saw": [d.seen for d in built if hasattr(d, "seen")], "public_state": [public_state(d) for d in built]}
    if cfg.get("mode") == "reuse":
        # the same manager on a second video (the first one backwards): what carries over, what clear() resets
        if cfg["clear_between"]:
            sm.clear()
        rev = frames[::-1]
        if cfg.get("other_size2"):      # the second video has another frame size (the reference's SAD detectors then refuse it without clear())
            rev = np.ascontiguousarray(rev[:, : max(1, rev.shape[1] - 4), : max(1, rev.shape[2] - 6)])
        video2 = (MemoryStream(rev, fps) if side in ("ref", "cross", "plug", "guest_cross") else psd.ArrayVideoStream(rev, fps))
        if "pts2" in cfg:      # the second video with presentation timestamps (whatever the first one had)
            video2 = (VfrMemoryStream(rev, fps, cfg["pts2"], Fraction(1, 1000)) if side in ("ref", "cross", "plug", "guest_cross")
                      else psd.ArrayVideoStream(rev, fps, pts=cfg["pts2"], time_base=Fraction(1, 1000)))
        n2, cuts2, scenes2, seen2 = detect(sm, video2)
        out["second"] = {"frames_processed": n2, "cuts": cuts2, "scenes": scenes2, "metrics": {}, "callback": seen2}
        out["num_detectors"] = sm.get_num_detectors()
        return out
    if cfg.get("second_pass") and stats is not None:
        import tempfile

        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "stats.csv")
            stats.save_to_csv(path)
            with open(path) as f:
                out["csv"] = f.read()
            if side in ("ref", "plug"):
                stats2 = RefStatsManager()
                sm2 = RefSceneManager(stats2)
            else:
                stats2 = psd.StatsManager()
                sm2 = psd.SceneManager(stats2, engine=engine, batch_frames=int(cfg.get("batch_frames", 64)))
            out["loaded"] = stats2.load_from_csv(path)
            if "second_dets" in cfg:
                dets = cfg["second_dets"]          # (configure() reads `dets`)
            built2 = configure(sm2)
            video.reset()
            n2, cuts2, scenes2, seen2 = detect(sm2, video)
            out["second"] = {"frames_processed": n2, "cuts": cuts2, "scenes": scenes2, "metrics": metrics_of(stats2, built2), "callback": seen2}
    return out


def public_state(detector):
    """The detector's public instance attributes after the run (reference: ThresholdDetector's ``last_fade`` / ``last_scene_cut`` /
    ``processed_frame`` ..., AdaptiveDetector's parameters), timecodes as frame numbers."""
    from enum import Enum

    def plain(v):
        if hasattr(v, "frame_num") and hasattr(v, "frame_rate"):
            return ["timecode", int(v.frame_num)]
        if isinstance(v, dict):
            return {str(k): plain(x) for k, x in sorted(v.items())}
        if isinstance(v, Enum):
            return ["enum", v.name]
        if hasattr(v, "pts") and hasattr(v, "time_base"):
            return ["pts", int(v.pts), str(v.time_base)]
        return [type(v).__name__, repr(v)]

    return {k: plain(v) for k, v in sorted(vars(detector).items()) if not k.startswith("_") and k not in ("seen", "last", "jump", "behind")}


def decisions(out):
    """An outcome without what only a full run has (warnings, log records): for sides that replay decisions from records."""
    return {k: v for k, v in out.items() if k not in ("warnings", "log")}


def sim_engine(oracle):
    """A stand-in of the DEVICE engine in host memory (tests/test_feed_rows.py: buffers that start poisoned, batched row uploads that
    land only at the fence) with the rest of what SceneManager, the feeder and the per-frame scorer ask of ``ScoringEngine``: the
    Python half of the GPU path -- device feeder, tap rows, slots and halo, frames wanted or pending, the resident per-frame path --
    runs on CPU under the same fuzz."""
    from tests.test_feed_rows import _AsyncHostBuffer, _BatchingEngine

    class Buffer(_AsyncHostBuffer):
        def upload(self, host, offset=0):
            host = np.ascontiguousarray(host)
            self.mem[offset: offset + host.nbytes] = host.reshape(-1).view(np.uint8)

        def download(self, nbytes=None, offset=0):
            nbytes = self.nbytes - offset if nbytes is None else nbytes
            return self.mem[offset: offset + nbytes].copy()

    class Sim(_BatchingEngine):
        def alloc(self, nbytes):
            self.buffers.append(Buffer(nbytes, self))
            return self.buffers[-1]

        def analyze_device(self, d_frames, n, height, width, frame_stride, d_prev=None, flags=0, edge_kernels=(0,), downscale=1.0,
                           hash_sizes=(), interpolation=1, want_frames=False):
            out = super().analyze_device(d_frames, n, height, width, frame_stride, d_prev=d_prev, flags=flags, edge_kernels=edge_kernels,
                                         downscale=downscale, hash_sizes=hash_sizes, interpolation=interpolation, want_frames=want_frames)
            if flags & 8:      # (the stand-in of the tests scores the first dilation size only; ScoringEngine.analyze_device: every one)
                frames = self._view(d_frames, n, height, width, frame_stride)
                prev = self._view(d_prev, 1, height, width, frame_stride)[0] if d_prev else None
                kw = {"downscale": downscale, "interpolation": interpolation} if downscale > 1.0 else {}
                for k in list(dict.fromkeys(edge_kernels))[1:]:
                    out["edge_xor"][k] = self.oracle.score_host(frames, prev=prev, flags=8, edge_kernel=k, **kw)["edge_xor"]
            return out

        def score_host(self, *a, **k):
            return self.oracle.score_host(*a, **k)

        def hash_thumbs_host(self, *a, **k):
            return self.oracle.hash_thumbs_host(*a, **k)

        def downscale_host(self, *a, **k):
            return self.oracle.downscale_host(*a, **k)

        def score_device(self, d_frames, n, height, width, row_stride=None, frame_stride=None, d_prev=None, flags=0, edge_kernel=0, **_):
            stride = height * width * 3 if frame_stride is None else frame_stride
            frames = self._view(d_frames, n, height, width, stride)
            prev = self._view(d_prev, 1, height, width, stride)[0] if d_prev else None
            return self.oracle.score_host(frames, prev=prev, flags=flags, edge_kernel=edge_kernel)

    return Sim(oracle)


def outcome(fn):
    """What running one side gave: its results or its exception (type and text; the reference's bare asserts have no text), and the
    warnings it emitted on the way (category and text: deprecations, numpy's RuntimeWarnings) and everything it logged."""
    import logging
    import warnings

    class Capture(logging.Handler):
        def __init__(self):
            super().__init__(logging.DEBUG)
            self.records = []

        def emit(self, record):
            self.records.append([record.levelname, record.getMessage()])

    log, capture = logging.getLogger("pyscenedetect"), Capture()      # (the reference's logger; the mirror logs to the same name)
    saved = (log.level, log.propagate, logging.root.manager.disable)
    log.addHandler(capture)
    log.setLevel(logging.DEBUG)
    log.propagate = False
    logging.disable(logging.NOTSET)
    try:
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            try:
                out = fn()
            except Exception as ex:  # noqa: BLE001 -- the exception type IS the outcome
                out = {"raises": type(ex).__name__, "message": "" if isinstance(ex, AssertionError) else str(ex)}
    finally:
        log.removeHandler(capture)
        log.setLevel(saved[0])
        log.propagate = saved[1]
        logging.disable(saved[2])
    out["log"] = sorted(capture.records)    # everything that was logged (two threads log: the order between them is not an outcome)
    # (not compared: ResourceWarnings, and the logging module's own complaint about the reference's `logger.warn(...)` call)
    out["warnings"] = sorted({(w.category.__name__, str(w.message)) for w in caught if not issubclass(w.category, ResourceWarning)
                              and "'warn' method is deprecated" not in str(w.message)})
    return out


def differ(a, b, cfg=None):
    if cfg is not None and cfg.get("odd_frames") and (cfg.get("fail_at") or "raises" in a or "raises" in b):
        # a run that the caller's callback, the stream's read or a detector's exception ENDS: how far the decode thread had read ahead by then (and logged about
        # the frames it met) is a matter of timing in the reference (its queue holds four frames) and of the batch size here
        a, b = {k: v for k, v in a.items() if k != "log"}, {k: v for k, v in b.items() if k != "log"}
    if a.keys() != b.keys():
        return "outcome kinds: %s vs %s" % (sorted(a), sorted(b))
    if a.get("log") != b.get("log"):
        return "log: %s vs %s" % (str(a.get("log"))[:240], str(b.get("log"))[:240])
    if a.get("warnings") != b.get("warnings"):
        return "warnings: %s vs %s" % (str(a.get("warnings"))[:200], str(b.get("warnings"))[:200])
    if "raises" in a:
        if a["raises"] != b["raises"]:
            return "raises %s vs %s" % (a["raises"], b["raises"])
        return None if a.get("message") == b.get("message") else "%s says %r vs %r" % (a["raises"], a.get("message"), b.get("message"))
    for k in ("frames_processed", "cuts", "scenes", "callback", "csv", "loaded", "event_buffer", "num_detectors", "pieces", "stream", "plugin_saw", "public_state", "by_hand"):
        if a.get(k) != b.get(k):
            return "%s: %s vs %s" % (k, str(a.get(k))[:160], str(b.get(k))[:160])
    if ("second" in a) != ("second" in b):
        return "second pass on one side only"
    if "second" in a:
        why = differ(a["second"], b["second"])
        if why is not None:
            return "second pass (metrics from the loaded CSV): " + why
    if "metrics" not in a:
        return None
    if set(a["metrics"]) != set(b["metrics"]):
        return "metric keys: %s vs %s" % (sorted(a["metrics"]), sorted(b["metrics"]))
    for key, va in a["metrics"].items():
        vb = b["metrics"][key]
        for i, (x, y) in enumerate(zip(va, vb)):
            if x != y and not (x is not None and y is not None and x != x and y != y):
                return "%s[%d]: %r vs %r" % (key, i, x, y)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--force", default="", help="comma list of case features to force: per_frame, reuse, pts, stats, second_pass, callback")
    ap.add_argument("--wide", action="store_true", help="wider argument ranges, detection from a seek position, detection in pieces")
    ap.add_argument("--tiny", action="store_true", help="frames of a few pixels, clips of hundreds of frames: decision logic per second")
    ap.add_argument("--engines", action="store_true",
                    help="GPU box: the mirror over the HIP engine against the mirror over the oracle engine (no reference needed) -- the device "
                         "feeder, tap-row uploads, crop / downscale modes, batch sizes, the carried frame, the per-frame resident path")
    ap.add_argument("--cross", action="store_true",
                    help="third side: the mirror's SceneManager and detectors reading one of the REFERENCE's VideoStream objects (a user who "
                         "keeps the reference's decoder backend and swaps the rest)")
    ap.add_argument("--guest", action="store_true",
                    help="further sides, for cases that hold the plug-in detector: that detector derived from the REFERENCE's ABC (no extension "
                         "methods of this package) registered with the mirror's SceneManager, over the mirror's stream and over the reference's")
    ap.add_argument("--sim", action="store_true",
                    help="further side: the mirror over a host-memory stand-in of the DEVICE engine (feeder, tap rows, slots, halo, resident "
                         "per-frame path): the Python half of the GPU path on CPU")
    ap.add_argument("--plug", action="store_true",
                    help="third side: the mirror's detectors registered with the REFERENCE's SceneManager (its stream, its StatsManager, "
                         "its timecodes): the plug-in API as the reference itself drives it")
    ap.add_argument("--binding", action="store_true",
                    help="third side: the reference with INTEGRATION.md B's seams bound to the C-ABI (integration/scenedetect_amd.py over "
                         "oracle/libpsd_oracle_abi.so, the CPU build of the ABI), compared with the plain reference")
    args = ap.parse_args()
    import logging
    import warnings

    global TINY, WIDE
    TINY, WIDE = args.tiny, args.wide
    logging.disable(logging.CRITICAL)
    warnings.simplefilter("ignore")
    engine = OracleEngine()
    binding = amd = hip = None
    if args.engines:
        from pyscenedetect_amd.engine import ScoringEngine

        hip = ScoringEngine(0)
    elif ref is None:
        raise SystemExit("the reference checkout is not here: only --engines works on this box")
    if args.binding:
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libpsd_oracle_abi.so"])
        sys.path.insert(0, os.path.join(ROOT, "integration"))
        import scenedetect_amd as amd

        binding = amd.Binding(os.path.join(ROOT, "oracle", "libpsd_oracle_abi.so"))
    t_end = time.time() + args.seconds
    cases, by, raised, bad = 0, {}, 0, []
    while time.time() < t_end and cases < args.max_cases:
        rng = np.random.default_rng([args.seed, cases])
        frames, fps, dets, cfg = draw_case(rng)
        cfg["batch_frames"] = int(rng.choice([1, 7, 64]))
        for f in [x for x in args.force.split(",") if x]:
            if f in ("per_frame", "reuse", "mixed"):
                cfg["mode"] = f
                cfg.setdefault("clear_between", bool(cases & 1))
                cfg.setdefault("manager_first", bool(cases & 2))
                if f == "mixed" and cases % 3 == 0:
                    cfg["two_managers"] = True
                if f == "mixed":
                    cfg.pop("pts", None)
                    if cases % 2 == 0:
                        cfg["mixed_downscale"] = True
            elif f == "pts" and "pts" not in cfg:
                steps = rng.choice([20, 40, 40, 40, 40, 60, 80], size=len(frames) - 1)
                cfg["pts"] = [0] + [int(x) for x in np.cumsum(steps)]
            elif f in ("stats", "second_pass", "callback"):
                cfg[f] = True
                if f == "second_pass":
                    cfg["stats"] = True
                if cfg.get("stats"):
                    cfg.pop("frame_skip", None)
        if args.engines:
            a = outcome(lambda: run_side("mirror", frames, fps, dets, cfg, engine))
            b = outcome(lambda: run_side("mirror", frames, fps, dets, cfg, hip))
        else:
            a = outcome(lambda: run_side("ref", frames, fps, dets, cfg, None))
            b = outcome(lambda: run_side("mirror", frames, fps, dets, cfg, engine))
        why = differ(a, b, cfg)
        if why is None and binding is not None:
            undo = amd.install(binding)
            try:
                c = outcome(lambda: run_side("ref", frames, fps, dets, cfg, None))
            finally:
                undo()
            why = differ(a, c, cfg)
            if why is not None:
                why = "reference with the bound seams vs plain reference: " + why
        if why is None and args.cross and not args.engines:
            c = outcome(lambda: run_side("cross", frames, fps, dets, cfg, engine))
            why = differ(a, c, cfg)
            if why is not None:
                why = "mirror over the reference's stream vs plain reference: " + why
        if why is None and args.guest and not args.engines and any(name == "MeanJump" for name, _ in dets):
            for side in ("guest", "guest_cross"):
                c = outcome(lambda: run_side(side, frames, fps, dets, cfg, engine))
                why = differ(a, c, cfg)
                if why is not None:
                    why = "a detector on the reference's ABC under the mirror's manager (%s) vs plain reference: %s" % (side, why)
                    break
        if why is None and args.sim and not args.engines:
            c = outcome(lambda: run_side("mirror", frames, fps, dets, cfg, sim_engine(engine)))
            why = differ(a, c, cfg)
            if why is not None:
                why = "mirror over the simulated device engine vs plain reference: " + why
        if why is None and args.plug and not args.engines:
            c = outcome(lambda: run_side("plug", frames, fps, dets, cfg, engine))
            why = differ(a, c, cfg)
            if why is not None:
                why = "mirror's detectors under the reference's manager vs plain reference: " + why
        for name, _ in dets:
            by[name] = by.get(name, 0) + 1
        raised += "raises" in a
        if why is not None:
            desc = {"case": cases, "why": why, "shape": list(frames.shape), "fps": fps, "detectors": dets, "scene_manager": cfg}
            bad.append(desc)
            if args.verbose:
                print(json.dumps(desc), flush=True)
            if len(bad) >= 20:
                break
        cases += 1
    print(json.dumps({"seed": args.seed, "cases": cases, "by_detector": by, "cases_that_raise_on_both_sides": raised,
                      "mismatches": bad[:20]}))


if __name__ == "__main__":
    main()
