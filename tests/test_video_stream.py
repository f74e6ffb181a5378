"""CPU: the `VideoStream` interface of the mirror (reference scenedetect/video_stream.py:43-222): same member names, the same
abstract set and the same exceptions as the reference's class (compared with the reference where it is present), a decoder
written against it works under `SceneManager`, and `ArrayVideoStream` fills every member."""
import inspect
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import video_stream as vs
from tests.conftest import golden_clip

REFERENCE = "/root/reference"


class CountingStream(psd.VideoStream):
    """What a user's backend looks like: frames come out of a callable, one at a time; no seeking."""

    BACKEND_NAME = "counting"

    def __init__(self, frames, fps=Fraction(30000, 1001)):
        self._frames, self._fps, self._read = frames, Fraction(fps), 0

    path = property(lambda self: "/dev/null/counting")
    name = property(lambda self: "counting")
    is_seekable = property(lambda self: False)
    frame_rate = property(lambda self: self._fps)
    duration = property(lambda self: None)                      # "non terminating" as far as the stream knows
    frame_size = property(lambda self: (self._frames.shape[2], self._frames.shape[1]))
    aspect_ratio = property(lambda self: 1.0)
    frame_number = property(lambda self: self._read)
    position = property(lambda self: self.base_timecode + max(0, self._read - 1))
    position_ms = property(lambda self: self.position.seconds * 1000.0)

    def read(self, decode=True):
        if self._read >= len(self._frames):
            return False
        self._read += 1
        return self._frames[self._read - 1] if decode else True

    def reset(self):
        self._read = 0

    def seek(self, target):
        raise psd.SeekError("a counting stream cannot seek")


def test_interface_cannot_be_instantiated_and_names_its_abstract_members():
    with pytest.raises(TypeError):
        psd.VideoStream()
    assert vs.VideoStream.__abstractmethods__ == frozenset({
        "path", "name", "is_seekable", "frame_rate", "duration", "frame_size", "aspect_ratio", "position", "position_ms",
        "frame_number", "read", "reset", "seek"})
    assert issubclass(psd.FrameRateUnavailable, psd.VideoOpenFailure) and issubclass(psd.VideoOpenFailure, Exception)
    assert "framerate" in str(psd.FrameRateUnavailable()) and str(psd.VideoOpenFailure()) == "Unknown backend error."
    assert issubclass(psd.SeekError, Exception)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference exists in the build container only")
def test_interface_equals_the_reference_interface():
    """Public members, abstract set, signatures of the three methods and the exception messages, against the reference's own
    module (imported over the oracle's cv2 shim)."""
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "cv2_shim")
    added = [p for p in (shim, REFERENCE) if p not in sys.path]
    sys.path[:0] = added
    try:
        from scenedetect import video_stream as ref
    finally:
        for p in added:
            sys.path.remove(p)
    public = lambda cls: {n for n in dir(cls) if not n.startswith("_")}
    assert public(vs.VideoStream) == public(ref.VideoStream)
    assert vs.VideoStream.__abstractmethods__ == ref.VideoStream.__abstractmethods__
    for m in ("read", "reset", "seek"):
        assert list(inspect.signature(getattr(vs.VideoStream, m)).parameters) == list(inspect.signature(getattr(ref.VideoStream, m)).parameters), m
    assert inspect.signature(vs.VideoStream.read).parameters["decode"].default is True
    assert str(vs.FrameRateUnavailable()) == str(ref.FrameRateUnavailable()) and str(vs.VideoOpenFailure()) == str(ref.VideoOpenFailure())
    assert [c.__name__ for c in vs.FrameRateUnavailable.__mro__[:3]] == [c.__name__ for c in ref.FrameRateUnavailable.__mro__[:3]]
    # the members ArrayVideoStream adds are its own; everything the interface names it provides
    assert public(ref.VideoStream) <= public(vs.ArrayVideoStream)


def test_a_backend_written_against_the_interface_runs_under_scene_manager(golden, oracle_engine):
    frames = golden_clip(golden, "scenes_a")
    want = psd.SceneManager(engine=oracle_engine)
    want.add_detector(psd.ContentDetector(engine=oracle_engine))
    want.detect_scenes(psd.ArrayVideoStream(frames, Fraction(30000, 1001)))
    got = psd.SceneManager(engine=oracle_engine)
    got.add_detector(psd.ContentDetector(engine=oracle_engine))
    stream = CountingStream(frames)
    assert got.detect_scenes(stream) == len(frames) and stream.frame_number == len(frames)
    assert [c.frame_num for c in got.get_cut_list(show_warning=False)] == [c.frame_num for c in want.get_cut_list(show_warning=False)]
    assert len(got.get_cut_list(show_warning=False)) >= 2                                   # (the clip has cuts to find)
    assert [(a.frame_num, b.frame_num) for a, b in got.get_scene_list()] == [(a.frame_num, b.frame_num) for a, b in want.get_scene_list()]
    assert stream.decode_failures == 0 and stream.base_timecode.frame_rate == Fraction(30000, 1001)
    with pytest.raises(psd.SeekError):
        stream.seek(3)


def test_array_stream_fills_every_member():
    frames = np.zeros((5, 6, 8, 3), np.uint8)
    v = psd.ArrayVideoStream(frames, 25.0, name="clip")
    assert isinstance(v, psd.VideoStream) and v.BACKEND_NAME == "array"
    assert (v.path, v.name, v.is_seekable, v.aspect_ratio, v.decode_failures) == ("clip", "clip", True, 1.0, 0)
    assert v.frame_size == (8, 6) and v.frame_rate == 25 and v.duration.frame_num == 5
    assert v.frame_number == 0 and v.position.frame_num == 0 and v.position_ms == 0.0
    assert v.read() is not False and v.position_ms == 0.0 and v.frame_number == 1          # the first frame is shown at time 0
    assert v.read(decode=False) is True and v.position_ms == pytest.approx(40.0) and v.frame_number == 2
    v.seek(4)
    assert v.read() is not False and v.frame_number == 5 and v.read() is False
    v.reset()
    assert v.frame_number == 0
    with pytest.raises(ValueError):
        v.seek(-1)
    v.name = "renamed"
    assert v.name == "renamed" and v.path == "renamed"
    # presentation timestamps: position_ms follows them
    p = psd.ArrayVideoStream(frames, 25.0, pts=[0, 3003, 6006, 9009, 12012], time_base=Fraction(1, 90000))
    p.read(), p.read()
    assert p.position_ms == pytest.approx(3003 / 90.0)


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference exists in the build container only")
def test_top_level_names_of_the_reference_that_concern_the_path_exist_here():
    """`from scenedetect import X` -> `from pyscenedetect_amd import X` for every name of the reference's top level that belongs to
    the scoring path (scenedetect/__init__.py:33-78); decoders, output writers and the CLI are out of scope."""
    import ast

    tree = ast.parse(open(os.path.join(REFERENCE, "scenedetect", "__init__.py")).read())
    exported = {a.asname or a.name for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and n.module and n.module.startswith("scenedetect")
                for a in n.names}
    exported |= {n.name for n in tree.body if isinstance(n, ast.FunctionDef)}
    out_of_scope = {
        # platform / logging helpers, decoders, output writers
        "init_logger", "StrPath", "get_and_create_path", "open_video", "AVAILABLE_BACKENDS", "VideoStreamCv2", "VideoStreamAv", "VideoStreamMoviePy",
        "VideoCaptureAdapter", "VideoStreamConcat", "SourceSpan", "save_images", "split_video_ffmpeg", "split_video_mkvmerge",
        "is_ffmpeg_available", "is_mkvmerge_available", "write_scene_list", "write_scene_list_html", "PathFormatter", "VideoMetadata",
        "SceneMetadata",
    }
    missing = sorted(n for n in exported - out_of_scope if not hasattr(psd, n))
    assert not missing, missing
