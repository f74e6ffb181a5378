"""Writes tests/golden/api_signatures.json: public members and call signatures of the reference's classes on the scoring path
(evaluated from the unmodified reference over the cv2 shim), for tests/test_api_signatures.py.  Test infrastructure.
usage: PYTHONPATH=oracle/cv2_shim:/root/reference python oracle/gen_signature_golden.py"""
import inspect
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle", "cv2_shim"), "/root/reference"]

import scenedetect as ref  # noqa: E402
from scenedetect import detector as rdet, scene_manager as rsm, video_stream as rvs  # noqa: E402

CLASSES = {
    "ContentDetector": ref.ContentDetector, "AdaptiveDetector": ref.AdaptiveDetector, "ThresholdDetector": ref.ThresholdDetector,
    "HistogramDetector": ref.HistogramDetector, "HashDetector": ref.HashDetector, "SceneDetector": ref.SceneDetector,
    "FlashFilter": rdet.FlashFilter, "SceneManager": ref.SceneManager, "StatsManager": ref.StatsManager,
    "FrameTimecode": ref.FrameTimecode, "VideoStream": rvs.VideoStream,
}
FUNCTIONS = {"compute_downscale_factor": rsm.compute_downscale_factor, "get_scenes_from_cuts": rsm.get_scenes_from_cuts,
             "expand_scenes_to_bounds": rsm.expand_scenes_to_bounds, "detect": ref.detect}


def params(fn):
    return [[p.name, "<required>" if p.default is inspect.Parameter.empty else repr(p.default), p.kind.name]
            for p in inspect.signature(fn).parameters.values()]


out = {"reference_version": ref.__version__, "classes": {}, "functions": {k: params(v) for k, v in FUNCTIONS.items()}}
for name, cls in CLASSES.items():
    members = sorted(n for n in dir(cls) if not n.startswith("_"))
    sigs = {}
    for n in members + ["__init__"]:
        attr = inspect.getattr_static(cls, n)
        fn = attr.__func__ if isinstance(attr, (staticmethod, classmethod)) else attr
        if inspect.isfunction(fn):
            sigs[n] = params(fn)
    out["classes"][name] = {"members": members, "signatures": sigs,
                            "properties": sorted(n for n in members if isinstance(inspect.getattr_static(cls, n), property))}
path = os.path.join(ROOT, "tests", "golden", "api_signatures.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(path, sum(len(c["signatures"]) for c in out["classes"].values()), "signatures")
