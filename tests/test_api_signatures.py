"""CPU: every public member of the reference's classes on the scoring path exists on the mirror's class under the same name, as
the same kind of thing (property / method), and methods take the reference's parameters in the reference's order with the
reference's defaults (tests/golden/api_signatures.json, evaluated from the unmodified reference by oracle/gen_signature_golden.py).
The mirror may ADD keyword parameters behind them (`engine=`, `batch_frames=`, `base_timecode=`)."""
import inspect
import json
import os

import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import detector, scene_manager, video_stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")) as f:
    GOLD = json.load(f)
MIRROR = {"FlashFilter": detector.FlashFilter, "VideoStream": video_stream.VideoStream}


def mirror_class(name):
    return MIRROR.get(name) or getattr(psd, name)


def same_default(got: str, want: str) -> bool:
    # equal as written, or equal where only the spelling of the owner differs (enum members: <Mode.MERGE: 0> in both)
    return got == want or got.split(".")[-1] == want.split(".")[-1]


@pytest.mark.parametrize("name", sorted(GOLD["classes"]))
def test_class_surface(name):
    gold, cls = GOLD["classes"][name], mirror_class(name)
    missing = [m for m in gold["members"] if not hasattr(cls, m)]
    assert not missing, (name, missing)
    for p in gold["properties"]:
        assert isinstance(inspect.getattr_static(cls, p), property), (name, p, "is a property in the reference")
    for meth, want in gold["signatures"].items():
        attr = inspect.getattr_static(cls, meth)
        fn = attr.__func__ if isinstance(attr, (staticmethod, classmethod)) else attr
        assert inspect.isfunction(fn), (name, meth)
        got = [[p.name, "<required>" if p.default is inspect.Parameter.empty else repr(p.default), p.kind.name]
               for p in inspect.signature(fn).parameters.values()]
        assert [g[0] for g in got[:len(want)]] == [w[0] for w in want], (name, meth, got, want)
        for g, w in zip(got, want):
            assert g[2] == w[2], (name, meth, w[0], "kind", g[2], w[2])
            assert same_default(g[1], w[1]), (name, meth, w[0], g[1], w[1])
        for extra in got[len(want):]:
            assert extra[1] != "<required>" or extra[2] in ("VAR_POSITIONAL", "VAR_KEYWORD"), (name, meth, "the mirror adds a REQUIRED parameter", extra)


@pytest.mark.parametrize("name", sorted(GOLD["functions"]))
def test_function_signatures(name):
    fn = getattr(scene_manager, name, None) or getattr(psd, name)
    want = GOLD["functions"][name]
    got = [[p.name, "<required>" if p.default is inspect.Parameter.empty else repr(p.default)] for p in inspect.signature(fn).parameters.values()]
    # (detect(): the reference's parameters in its order -- `video_path` also takes decoded frames or a stream, a path goes to the
    #  reference's decoders where installed -- followed by this package's `fps` and `engine`)
    assert [g[0] for g in got[:len(want)]] == [w[0] for w in want], (name, got, want)
    for g, w in zip(got, want):
        assert same_default(g[1], w[1]), (name, w[0], g[1], w[1])


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/scenedetect"), reason="the reference checkout is only in the build container")
def test_module_paths_of_the_hot_path_carry_over():
    """An import written for the reference needs only the package name changed: every module of the hot path exists under the same path
    (``scenedetect.common`` -> ``pyscenedetect_amd.common`` ...) and holds every public name the reference defines there (SURVEY.md 2,
    rows 1-8 and 16; output, decoding and CLI modules are out of scope)."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    added = [p for p in (os.path.join(root, "oracle", "cv2_shim"), "/root/reference") if p not in sys.path]
    sys.path[:0] = added
    try:
        out_of_scope = {"scene_manager": {"save_images", "write_scene_list", "write_scene_list_html", "PathFormatter", "SceneMetadata",
                                          "VideoMetadata", "default_formatter", "is_ffmpeg_available", "is_mkvmerge_available",
                                          "split_video_ffmpeg", "split_video_mkvmerge", "write_scene_list_edl", "write_scene_list_fcp7",
                                          "write_scene_list_fcpx", "write_scene_list_otio"}}
        for path in ("common", "detector", "detectors", "detectors.content_detector", "detectors.adaptive_detector",
                     "detectors.histogram_detector", "detectors.threshold_detector", "detectors.hash_detector", "scene_manager",
                     "stats_manager", "video_stream"):
            theirs = importlib.import_module("scenedetect." + path)
            ours = importlib.import_module("pyscenedetect_amd." + path)
            defined = [n for n in dir(theirs) if not n.startswith("_")
                       and getattr(getattr(theirs, n), "__module__", None) == theirs.__name__]
            defined += list(getattr(theirs, "__all__", []))
            missing = sorted(set(defined) - set(dir(ours)) - out_of_scope.get(path, set()))
            assert not missing, (path, missing)
    finally:
        for p in added:
            sys.path.remove(p)
