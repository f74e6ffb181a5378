"""CPU (+ one GPU pass): SceneManager behaviour around the per-frame loop -- seek/end_time/duration
windows, start_in_scene, callbacks with look-behind, crop, frame_skip, stats CSV -- against runs of
the unmodified reference (tests/golden/reference_runs.json["scenarios"]; the reference's own
versions of these tests are tests/test_scene_manager.py:28-196 and need video files)."""
import io

import numpy as np
import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import FlashFilter
from tests.conftest import golden_clip


def _manager(det, engine, stats=False, **attrs):
    sm = psd.SceneManager(psd.StatsManager() if stats else None, engine=engine, batch_frames=32)
    sm.auto_downscale = False
    for k, v in attrs.items():
        setattr(sm, k, v)
    sm.add_detector(det)
    return sm


def _scenes(sm, **kw):
    return [[a.frame_num, b.frame_num] for a, b in sm.get_scene_list(**kw)]


def _run_scenarios(golden, engine):
    frames = golden_clip(golden, "scenes_a")
    want = golden["scenarios"]

    video = psd.ArrayVideoStream(frames, 25.0)
    video.seek(40)
    sm = _manager(psd.ContentDetector(engine=engine), engine)
    assert sm.detect_scenes(video, end_time=200) == want["window_seek40_end200"]["frames_processed"]
    assert _scenes(sm) == want["window_seek40_end200"]["scenes"]
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == want["window_seek40_end200"]["cuts"]
    scenes = sm.get_scene_list()
    assert all(a.frame_num < b.frame_num for a, b in scenes) and all(x[1] == y[0] for x, y in zip(scenes, scenes[1:]))

    sm = _manager(psd.ContentDetector(engine=engine), engine)
    assert sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), duration=100) == want["duration_100"]["frames_processed"]
    assert _scenes(sm) == want["duration_100"]["scenes"]

    sm = _manager(psd.ContentDetector(engine=engine), engine)
    assert sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), end_time=10) == want["short_no_cuts"]["frames_processed"]
    assert _scenes(sm) == [] and _scenes(sm, start_in_scene=True) == want["short_no_cuts"]["scenes_start_in_scene"]

    dets = {"content": lambda: psd.ContentDetector(engine=engine),
            "adaptive": lambda: psd.AdaptiveDetector(engine=engine),
            "content_suppress": lambda: psd.ContentDetector(filter_mode=FlashFilter.Mode.SUPPRESS, min_scene_len=6,
                                                            threshold=20.0, engine=engine)}
    for name, make in dets.items():
        calls = []
        sm = _manager(make(), engine)
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), callback=lambda img, pos: calls.append([pos.frame_num, int(img.sum())]))
        assert calls == want[f"callback_{name}"]["calls"], name
        assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == want[f"callback_{name}"]["cuts"]

    sm = _manager(psd.ContentDetector(engine=engine), engine, stats=True, crop=(10, 5, 100, 60))
    assert sm.crop == (10, 5, 100, 60)
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == want["crop_10_5_100_60"]["cuts"]
    cv = [sm.stats_manager.get_metrics(i, ["content_val"])[0] for i in range(len(frames))]
    assert [None if v is None else float(v) for v in cv] == want["crop_10_5_100_60"]["content_val"]

    sm = _manager(psd.ContentDetector(engine=engine), engine)
    assert sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), frame_skip=1) == want["frame_skip_1"]["frames_processed"]
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == want["frame_skip_1"]["cuts"]
    assert _scenes(sm) == want["frame_skip_1"]["scenes"]


def test_scenarios_match_reference(golden, oracle_engine):
    _run_scenarios(golden, oracle_engine)


@pytest.mark.gpu
def test_scenarios_match_reference_on_gpu(golden, hip_engine):
    _run_scenarios(golden, hip_engine)


def test_argument_validation(oracle_engine):
    sm = psd.SceneManager(psd.StatsManager(), engine=oracle_engine)
    sm.add_detector(psd.ThresholdDetector(engine=oracle_engine))
    video = psd.ArrayVideoStream(np.zeros((4, 8, 8, 3), np.uint8), 25.0)
    with pytest.raises(ValueError):
        sm.detect_scenes(video, frame_skip=1)              # not allowed with a StatsManager
    with pytest.raises(ValueError):
        sm.detect_scenes(video, duration=1, end_time=2)
    with pytest.raises(ValueError):
        sm.detect_scenes(video, end_time=-1)
    with pytest.raises(TypeError):
        sm.detect_scenes()
    with pytest.raises(ValueError):
        sm.crop = (0, 0, -1, 4)
    with pytest.raises(TypeError):
        sm.crop = (0, 0, 1.5, 4)
    sm2 = psd.SceneManager(engine=oracle_engine)
    sm2.crop = (100, 100, 120, 120)
    sm2.add_detector(psd.ThresholdDetector(engine=oracle_engine))
    with pytest.raises(ValueError, match="outside video boundary"):
        sm2.detect_scenes(psd.ArrayVideoStream(np.zeros((4, 8, 8, 3), np.uint8), 25.0))
    with pytest.raises(ValueError):
        psd.ContentDetector(kernel_size=4)
    with pytest.raises(ValueError):
        psd.AdaptiveDetector(window_width=0)
    with pytest.raises(ValueError):
        psd.HistogramDetector(engine=oracle_engine).process_frame(psd.FrameTimecode(0, 25.0), np.zeros((4, 4, 3), np.float32))


def test_decode_errors_surface_in_caller(oracle_engine):
    """An exception in the frame source is re-raised by detect_scenes after the worker stopped
    (reference scene_manager.py:598-618)."""

    class Broken(psd.ArrayVideoStream):
        def read(self, decode=True):
            if self.frame_number == 5:
                raise OSError("decoder blew up")
            return super().read(decode)

    sm = psd.SceneManager(engine=oracle_engine, batch_frames=2)
    sm.add_detector(psd.ThresholdDetector(engine=oracle_engine))
    with pytest.raises(OSError, match="decoder blew up"):
        sm.detect_scenes(Broken(np.zeros((9, 8, 8, 3), np.uint8), 25.0))


def test_stats_csv_and_cached_threshold_metric(golden, oracle_engine):
    frames = golden_clip(golden, "fades_b")
    stats = psd.StatsManager()
    sm = psd.SceneManager(stats, engine=oracle_engine)
    sm.auto_downscale = False
    sm.add_detector(psd.ThresholdDetector(engine=oracle_engine))
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    cuts = [c.frame_num for c in sm.get_cut_list(show_warning=False)]
    assert cuts == golden["clips"]["fades_b"]["results"]["threshold_default"]["cuts"]
    buf = io.StringIO()
    stats.save_to_csv(buf)
    lines = buf.getvalue().strip().split("\n")
    assert lines[0] == "Frame Number,Timecode,average_rgb" and len(lines) == len(frames) + 1
    assert lines[1].startswith("1,00:00:00.000,")
    # second pass over cached metrics: no pixel work at all (threshold_detector.py:122-125)
    class NoEngine:
        def score_host(self, *a, **k):
            raise AssertionError("cached average_rgb must be used")

    det = psd.ThresholdDetector(engine=NoEngine())
    det.stats_manager = stats
    again = []
    for i, f in enumerate(frames):
        again += det.process_frame(psd.FrameTimecode(i, 25.0), f)
    assert [c.frame_num for c in again] == cuts


def test_detect_convenience(golden, oracle_engine, tmp_path):
    """``detect()`` = open -> SceneManager -> scene list (reference scenedetect/__init__.py:160-219)."""
    frames = golden_clip(golden, "wide_d")           # > 256 px wide: default auto-downscale applies
    want = golden["clips"]["wide_d"]["results"]["content_default"]["scenes"]
    scenes = psd.detect(frames, psd.ContentDetector(engine=oracle_engine), engine=oracle_engine)
    assert [[a.frame_num, b.frame_num] for a, b in scenes] == want
    csv_path = tmp_path / "stats.csv"
    psd.detect(frames, psd.ThresholdDetector(engine=oracle_engine), stats_file_path=str(csv_path), engine=oracle_engine)
    assert csv_path.read_text().startswith("Frame Number,Timecode,average_rgb")
    empty = psd.detect(frames[:5], psd.ContentDetector(engine=oracle_engine), engine=oracle_engine)
    assert empty == [] and len(psd.detect(frames[:5], psd.ContentDetector(engine=oracle_engine), start_in_scene=True,
                                          engine=oracle_engine)) == 1


def test_stats_manager_csv_round_trip_and_corrupt_files(golden, oracle_engine, tmp_path):
    """save_to_csv -> load_from_csv (reference stats_manager.py:164-296): same metric values back, keyed by 0-based
    frame number; blank / missing files give None; anything else that is not a stats file raises StatsFileCorrupt."""
    frames = golden_clip(golden, "scenes_a")
    stats = psd.StatsManager()
    sm = psd.SceneManager(stats, engine=oracle_engine)
    sm.auto_downscale = False
    sm.add_detector(psd.ContentDetector(engine=oracle_engine))
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    path = tmp_path / "stats.csv"
    stats.save_to_csv(str(path))
    loaded = psd.StatsManager()
    assert loaded.load_from_csv(str(path)) == len(frames) - 1      # frame 0 has no metrics (nothing to compare with)
    assert not loaded.is_save_required()
    keys = ["content_val", "delta_hue", "delta_sat", "delta_lum"]
    for i in range(1, len(frames)):
        assert loaded.metrics_exist(i, keys)
        assert loaded.get_metrics(i, keys) == [float(str(v)) for v in stats.get_metrics(i, keys)]
    assert not loaded.metrics_exist(0, keys)
    assert set(keys) <= set(loaded.metric_keys)
    # blank and missing files
    blank = tmp_path / "blank.csv"
    blank.write_text("")
    assert psd.StatsManager().load_from_csv(str(blank)) is None
    assert psd.StatsManager().load_from_csv(str(tmp_path / "nope.csv")) is None
    # an older layout with one extra line in front of the header still loads
    legacy = tmp_path / "legacy.csv"
    legacy.write_text("Video Framerate,25.0\nFrame Number,Timecode,content_val\n1,00:00:00.000,3.5\n2,00:00:00.040,None\n")
    old = psd.StatsManager()
    assert old.load_from_csv(str(legacy)) == 2
    assert old.get_metrics(0, ["content_val"]) == [3.5] and not old.metrics_exist(1, ["content_val"])
    # corrupt files
    for text in ("a,b,c\n1,2,3\n",                                              # not a stats header
                 "Frame Number,Timecode\n1,00:00:00.000\n",                      # no metrics
                 "Frame Number,Timecode,content_val\n1,00:00:00.000\n",          # short row
                 "Frame Number,Timecode,content_val\n1,00:00:00.000,abc\n"):     # not a number
        bad = tmp_path / "bad.csv"
        bad.write_text(text)
        with pytest.raises(psd.StatsFileCorrupt):
            psd.StatsManager().load_from_csv(str(bad))


def test_crop_setter_validation(oracle_engine):
    """Same accept / reject behaviour as the reference's setter (scene_manager.py:298-321, tests/test_scene_manager.py:199)."""
    sm = psd.SceneManager(engine=oracle_engine)
    for ok in (None, (0, 0, 0, 0), (1, 1, 0, 0), (0, 0, 1, 1)):
        sm.crop = ok
    for bad in (1, (1, 1), (1, 1, 1)):
        with pytest.raises(TypeError):
            sm.crop = bad
    with pytest.raises(ValueError):
        sm.crop = (1, 1, 1, -1)


def test_expand_scenes_to_bounds():
    """Outer endpoints are replaced, inner boundaries kept, the input list untouched (reference scene_manager.py:143-168)."""
    tc = lambda n: psd.FrameTimecode(n, 10.0)  # noqa: E731
    scenes = [(tc(130), tc(150)), (tc(150), tc(170))]
    before = list(scenes)
    assert psd.expand_scenes_to_bounds(scenes, tc(0), tc(300)) == [(tc(0), tc(150)), (tc(150), tc(300))]
    assert scenes == before
    assert psd.expand_scenes_to_bounds([(tc(130), tc(170))], tc(0), tc(300)) == [(tc(0), tc(300))]
    assert psd.expand_scenes_to_bounds([], tc(0), tc(100)) == []


# ---- several detectors in one manager: nobody's results depend on who else is registered -------------------------------

class _ProbeDetector(psd.SceneDetector):
    """A plug-in detector without a device path: records what process_frame is handed."""

    def __init__(self):
        super().__init__()
        self.shapes = []
        self.sums = []

    def get_metrics(self):
        return []

    def process_frame(self, timecode, frame_img):
        self.shapes.append(frame_img.shape)
        self.sums.append(int(frame_img.sum()))
        return []


def _independence_checks(golden, engine):
    import cv2  # the oracle shim

    frames = golden_clip(golden, "wide_d")                       # 320 x 180: auto-downscale to 256 x 144

    def run(dets, callback=None, stats=True):
        sm = psd.SceneManager(psd.StatsManager() if stats else None, engine=engine, batch_frames=16)
        for d in dets:
            sm.add_detector(d)
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), callback=callback)
        return sm

    def metric(sm, key):
        return [sm.stats_manager.get_metrics(i, [key])[0] if sm.stats_manager.metrics_exist(i, [key]) else None for i in range(len(frames))]

    # plug-in detectors and callbacks see the DOWNSCALED frame, like the reference (its decode thread resizes first)
    probe = _ProbeDetector()
    seen = []
    sm = run([psd.ContentDetector(engine=engine), probe], callback=lambda img, pos: seen.append((pos.frame_num, img.shape, int(img.sum()))))
    small = [cv2.resize(f, (256, 144)) for f in frames]
    assert set(probe.shapes) == {(144, 256, 3)} and probe.sums == [int(s.sum()) for s in small]
    assert seen and all(shape == (144, 256, 3) and total == int(small[n].sum()) for n, shape, total in seen)
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == golden["clips"]["wide_d"]["results"]["content_default"]["cuts"]

    # two thumbnail sizes in one pass: each HashDetector's metrics equal its solo run (and the reference's)
    key8, key16 = "hash_dist [size=8 lowpass=2]", "hash_dist [size=16 lowpass=2]"
    solo8 = metric(run([psd.HashDetector(engine=engine)]), key8)
    solo16 = metric(run([psd.HashDetector(size=16, lowpass=2, threshold=0.3, min_scene_len=5, engine=engine)]), key16)
    both = run([psd.HashDetector(engine=engine), psd.HashDetector(size=16, lowpass=2, threshold=0.3, min_scene_len=5, engine=engine),
                _ProbeDetector()])
    assert metric(both, key8) == solo8 == golden["clips"]["wide_d"]["results"]["hash_default"]["metrics"][key8]
    assert metric(both, key16) == solo16 == golden["clips"]["wide_d"]["results"]["hash_16_lp2"]["metrics"][key16]

    # two edge-dilation sizes in one pass: each detector is scored with its own kernel (content_detector.py:135-137)
    class Scored(psd.ContentDetector):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.scores = []

        def process_record(self, timecode, record, height, width):
            cuts = super().process_record(timecode, record, height, width)
            self.scores.append(self._frame_score)
            return cuts

    frames = frames.copy()
    for t in range(len(frames)):                                   # something with edges that moves
        frames[t, 40 + t:90 + t, 60 + 2 * t:140 + 2 * t] = (230, 40, 200)
    w = psd.ContentDetector.Components(0.0, 0.0, 0.0, 1.0)
    mk = lambda k: Scored(weights=w, kernel_size=k, engine=engine)  # noqa: E731
    solo = []
    for k in (3, 9, None):
        d = mk(k)
        run([d], stats=False)
        solo.append(d.scores)
    assert solo[0] != solo[1] and solo[1] != solo[2] and any(solo[0])
    for order in ((0, 1, 2), (2, 1, 0)):
        dets = [mk((3, 9, None)[i]) for i in order]
        run(dets, stats=False)
        for i, d in zip(order, dets):
            assert d.scores == solo[i], f"kernel {(3, 9, None)[i]} scored differently next to other detectors"


def test_detectors_do_not_depend_on_co_registered_detectors(golden, oracle_engine):
    _independence_checks(golden, oracle_engine)


@pytest.mark.gpu
def test_detectors_do_not_depend_on_co_registered_detectors_gpu(golden, hip_engine):
    _independence_checks(golden, hip_engine)


def test_benchmark_harness_predictions_and_reference_evaluator(tmp_path, golden, oracle_engine):
    """tools/bbc_harness.py runs the reference's benchmark loop on this package; tools/bbc_evaluate.py scores the
    predictions with the reference's own evaluator (only where the reference checkout exists)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bbc_harness as H

    frames = golden_clip(golden, "scenes_a")
    os.makedirs(tmp_path / "videos")
    os.makedirs(tmp_path / "fixed")
    truth = golden["clips"]["scenes_a"]["true_cuts"]
    for vid in ("01", "02"):
        np.save(tmp_path / "videos" / f"bbc_{vid}.npy", frames)
        bounds = [0, *truth, len(frames)]
        with open(tmp_path / "fixed" / f"{vid}-scenes.txt", "w") as f:
            for a, b in zip(bounds[:-1], bounds[1:]):
                f.write(f"{a}\t{b - 1}\n")          # start, last frame of the scene (0-based), as in the BBC files
    samples = H.bbc_samples(str(tmp_path))
    assert [s["hard_cuts"] for s in samples] == [[*truth, len(frames)]] * 2
    recs = H.run_predictions(samples, "detect-content", engine=oracle_engine)
    want = golden["clips"]["scenes_a"]["results"]["content_default"]["cuts"]
    assert [r["predicted_cuts"] for r in recs] == [[*want, len(frames)]] * 2
    H.dump(recs, str(tmp_path / "pred.json"), detector="detect-content")
    if not os.path.isdir("/root/reference/benchmark"):
        pytest.skip("the reference checkout (benchmark/evaluator.py) is only present in the build container")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bbc_evaluate.py"), str(tmp_path / "pred.json"), "--tolerances", "0,2"],
                         capture_output=True, text=True, check=True)
    res = json.loads(out.stdout)["results"]
    assert res[0]["tolerance"] == 0 and res[0]["videos"] == 2 and 0.0 < res[0]["f1"] <= res[1]["f1"] <= 1.0
    assert res[0]["matched"] + res[0]["missed"] == 2 * (len(truth) + 1)


def test_array_video_stream_contract():
    """The part of the reference's VideoStream contract SceneManager relies on (video_stream.py:79-222): frame_number
    counts the frames read, position is the timecode of the last frame read (frame 0 before any read), read(decode=False)
    advances without handing out a frame, False at the end, seek / reset move the read position."""
    from fractions import Fraction

    frames = np.arange(5 * 4 * 6 * 3, dtype=np.uint8).reshape(5, 4, 6, 3)
    v = psd.ArrayVideoStream(frames, 25.0)
    assert v.frame_size == (6, 4) and v.frame_rate == 25 and v.base_timecode.frame_num == 0
    assert v.duration.frame_num == 5 and v.frame_number == 0 and v.position.frame_num == 0
    f0 = v.read()
    assert np.array_equal(f0, frames[0]) and v.frame_number == 1 and v.position.frame_num == 0
    assert v.read(decode=False) is True and v.frame_number == 2 and v.position.frame_num == 1
    assert np.array_equal(v.read(), frames[2]) and v.position.frame_num == 2
    v.seek(4)
    assert np.array_equal(v.read(), frames[4]) and v.frame_number == 5
    assert v.read() is False and v.read(decode=False) is False and v.frame_number == 5 and v.position.frame_num == 4
    v.seek(psd.FrameTimecode(1, 25.0))
    assert np.array_equal(v.read(), frames[1])
    v.seek(99)
    assert v.read() is False
    with pytest.raises(ValueError):
        v.seek(-1)
    v.reset()
    assert v.frame_number == 0 and np.array_equal(v.read(), frames[0])
    # a list of separately allocated frames works like a stacked array; an empty source reads nothing
    lst = psd.ArrayVideoStream([f.copy() for f in frames], Fraction(30000, 1001))
    assert lst.frame_size == (6, 4) and lst.frame_rate == Fraction(30000, 1001) and np.array_equal(lst.read(), frames[0])
    empty = psd.ArrayVideoStream(frames[:0])
    assert empty.read() is False and empty.duration.frame_num == 0
    # presentation timestamps: positions and the duration carry them
    pts = [0, 40, 100, 140, 220]
    p = psd.ArrayVideoStream(frames, 25.0, pts=pts, time_base=Fraction(1, 1000))
    assert p.position.pts == 0
    p.read(); p.read(); p.read()
    assert p.position.pts == 100 and abs(p.position.seconds - 0.1) < 1e-12 and p.duration.pts == 220
    with pytest.raises(ValueError):
        psd.ArrayVideoStream(frames, 25.0, pts=pts)
    with pytest.raises(ValueError):
        psd.ArrayVideoStream(frames, 25.0, pts=pts[:3], time_base=Fraction(1, 1000))


def test_deprecated_spellings_of_the_reference(golden, oracle_engine):
    """`get_cut_list()` warns like the reference's (scene_manager.py:716-744) unless told not to; `frame_source=` still names the
    video (scene_manager.py:487-494); `save_to_csv(csv_file, force_save)` takes `force_save` second (stats_manager.py:164-168) and
    the StatsManager accessors take `timecode=` (stats_manager.py:126-153)."""
    frames = golden_clip(golden, "scenes_a")
    stats = psd.StatsManager()
    sm = psd.SceneManager(stats, engine=oracle_engine)
    sm.add_detector(psd.ContentDetector(engine=oracle_engine))
    with pytest.warns(DeprecationWarning, match="frame_source"):
        assert sm.detect_scenes(frame_source=psd.ArrayVideoStream(frames, 25.0)) == len(frames)
    with pytest.warns(DeprecationWarning, match="get_cut_list"):
        cuts = sm.get_cut_list()
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert sm.get_cut_list(show_warning=False) == cuts and len(cuts) >= 2
    key = psd.ContentDetector.FRAME_SCORE_KEY
    assert stats.metrics_exist(timecode=cuts[0], metric_keys=[key]) and stats.metrics_exist(timecode=cuts[0].frame_num, metric_keys=[key])
    assert stats.get_metrics(timecode=cuts[0], metric_keys=[key]) == stats.get_metrics(cuts[0].frame_num, [key])
    stats.set_metrics(timecode=3, metric_kv_dict={"custom": 1.5})
    assert stats.get_metrics(3, ["custom"]) == [1.5]
    buf = io.StringIO()
    stats._metrics_updated = False
    stats.save_to_csv(buf, False)                       # nothing new to write, and the second positional argument is force_save
    assert buf.getvalue() == ""
    stats.save_to_csv(buf, True)
    assert buf.getvalue().startswith("Frame Number,Timecode,")


def test_show_progress_drives_a_tqdm_bar_like_the_reference(oracle_engine, monkeypatch):
    """``detect_scenes(show_progress=True)``: a bar over the stream's remaining frames (to ``end_time`` if that comes first),
    advanced once per frame, its description carrying the cut count (reference ``scene_manager.py:549-563, 585-603``)."""
    import pyscenedetect_amd as psd
    from pyscenedetect_amd import scene_manager as SM
    from pyscenedetect_amd.synth import make_clip

    bars = []

    class Bar:
        def __init__(self, **kw):
            self.kw, self.n, self.descs, self.closed = kw, 0, [], False
            bars.append(self)

        def update(self, n=1):
            self.n += n

        def set_description(self, desc=None, refresh=True):
            self.descs.append((desc, refresh))

        def close(self):
            self.closed = True

    monkeypatch.setattr(SM, "_tqdm", Bar)
    frames, cuts = make_clip(3, 90, 36, 64, shot_len=(20, 30))
    for kwargs, total in (({}, 90), ({"end_time": 50}, 50), ({"duration": 40}, 40)):
        sm = psd.SceneManager(engine=oracle_engine)
        sm.add_detector(psd.ContentDetector(engine=oracle_engine, min_scene_len=5))
        n = sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), show_progress=True, **kwargs)
        bar = bars[-1]
        found = len(sm.get_cut_list(show_warning=False))
        assert bar.kw["total"] == total and bar.kw["unit"] == "frames" and bar.kw["desc"] == SM.PROGRESS_BAR_DESCRIPTION % 0
        assert bar.n == n and bar.closed
        assert bar.descs[-1] == (SM.PROGRESS_BAR_DESCRIPTION % found, True)
        assert found and all(not refresh for _, refresh in bar.descs[:-1])
    # without the flag nothing is created
    sm = psd.SceneManager(engine=oracle_engine)
    sm.add_detector(psd.ContentDetector(engine=oracle_engine))
    before = len(bars)
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    assert len(bars) == before


def test_decode_thread_moves_onto_the_cpus_the_engine_names_and_nobody_else_does():
    """``_run_near_gpu`` (the first thing SceneManager's own decode thread does when a device feeder is attached) applies
    ``ScoringEngine.cpus_near_gpu()`` to the CALLING thread only; an engine without the method, an empty answer or a refused
    call leave it where it was."""
    import os
    import threading

    from pyscenedetect_amd.scene_manager import _run_near_gpu

    mine = os.sched_getaffinity(0)
    one = {min(mine)}

    class Near:
        def cpus_near_gpu(self):
            return sorted(one)

    class Nothing:
        def cpus_near_gpu(self):
            return []

    class Refused:
        def cpus_near_gpu(self):
            return [10 ** 6]          # no such CPU: sched_setaffinity raises, the thread stays

    seen = {}

    def body(name, engine):
        _run_near_gpu(engine)
        seen[name] = os.sched_getaffinity(0)

    for name, engine in (("near", Near()), ("nothing", Nothing()), ("refused", Refused()), ("plain", object())):
        t = threading.Thread(target=body, args=(name, engine))
        t.start()
        t.join()
    assert seen["near"] == one
    assert seen["nothing"] == mine and seen["refused"] == mine and seen["plain"] == mine
    assert os.sched_getaffinity(0) == mine


def test_no_thread_outlives_detect_scenes(golden, oracle_engine):
    """Thread hygiene (reference ``scene_manager.py:598-613``: "the decode thread must never be abandoned", and its test session's
    thread-leak report, ``tests/conftest.py:186-211``): whatever ends a ``detect_scenes`` call -- the end of the video, ``end_time``, an
    exception from the caller's callback or from the stream's ``read``, ``stop()`` from the callback -- no thread of the manager (the
    decode thread, with a device engine also the feeder's) is alive afterwards."""
    import threading

    frames = golden_clip(golden, "scenes_a")
    before = {t.ident for t in threading.enumerate()}

    class Failing(psd.ArrayVideoStream):
        def read(self, decode=True):
            if self.frame_number == 40:
                raise OSError("decoder failure")
            return super().read(decode)

    def run(video, **kwargs):
        sm = psd.SceneManager(psd.StatsManager(), engine=oracle_engine, batch_frames=7)
        sm.add_detector(psd.ContentDetector(engine=oracle_engine))
        sm.add_detector(psd.AdaptiveDetector(engine=oracle_engine))
        if kwargs.pop("stop", False):
            kwargs["callback"] = lambda img, pos: sm.stop()
        return sm.detect_scenes(video, **kwargs)

    def failing_callback(img, pos):
        raise KeyError("caller's callback failed")

    assert run(psd.ArrayVideoStream(frames, 25.0)) == len(frames)
    assert run(psd.ArrayVideoStream(frames, 25.0), end_time=50) == 50
    with pytest.raises(KeyError):
        run(psd.ArrayVideoStream(frames, 25.0), callback=failing_callback)
    with pytest.raises(OSError):
        run(Failing(frames, 25.0))
    assert run(psd.ArrayVideoStream(frames, 25.0), stop=True) <= len(frames)
    for _ in range(50):                     # (a thread that is finishing gets a moment)
        left = [t for t in threading.enumerate() if t.ident not in before and t.name != "tqdm_monitor"]
        if not left:
            break
        import time

        time.sleep(0.02)
    assert not left, [t.name for t in left]


def test_happy_path_logs_nothing_at_error_level(golden, oracle_engine, caplog):
    """The reference's test session fails any test that logs at ERROR level (``tests/conftest.py:91-101``); so do runs of the mirror:
    every detector, downscale, crop, callback and a StatsManager on a clean clip log no error (a frame of another size does, as in
    the reference, ``scene_manager.py:645-664``)."""
    import logging

    frames = golden_clip(golden, "wide_d")
    with caplog.at_level(logging.DEBUG):
        sm = psd.SceneManager(psd.StatsManager(), engine=oracle_engine, batch_frames=16)
        sm.crop = (3, 2, 300, 170)
        for cls in (psd.ContentDetector, psd.AdaptiveDetector, psd.HistogramDetector, psd.ThresholdDetector, psd.HashDetector):
            sm.add_detector(cls(engine=oracle_engine))
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), callback=lambda img, pos: None)
        assert sm.get_scene_list(start_in_scene=True)
    assert not [r for r in caplog.records if r.levelno >= logging.ERROR], [r.getMessage() for r in caplog.records]
    caplog.clear()
    odd = list(frames[:12])
    odd[5] = np.ascontiguousarray(odd[5][:-4, :-6])
    with caplog.at_level(logging.DEBUG):
        sm = psd.SceneManager(engine=oracle_engine)
        sm.add_detector(psd.ContentDetector(engine=oracle_engine))
        assert sm.detect_scenes(psd.ArrayVideoStream(odd, 25.0)) == 12
    assert len([r for r in caplog.records if r.levelno >= logging.ERROR]) == 1


def test_long_stream_keeps_memory_bounded(oracle_engine):
    """The reference's stress test (``tests/release/test_long_video.py:34-84``: RSS <= 3x baseline on a 15-minute clip) on a synthetic
    stream of 40 000 frames that exist one at a time: without a StatsManager the manager keeps the look-behind buffer and one batch,
    nothing that grows with the video (the reference's loop holds Queue(4) frames, ``scene_manager.py:113,422-425``)."""
    import gc

    psutil = pytest.importorskip("psutil")

    class Endless:
        """Frames made on demand: a shot every 97 frames, noise on top."""
        shape = (40000, 36, 64, 3)

        def __len__(self):
            return self.shape[0]

        def __getitem__(self, i):
            rng = np.random.default_rng(i)
            return (rng.integers(0, 24, (36, 64, 3)) + 13 * ((i // 97) % 17)).astype(np.uint8)

    def rss():
        gc.collect()
        return psutil.Process().memory_info().rss

    def run(n):
        frames = Endless()
        frames.shape = (n,) + frames.shape[1:]
        sm = psd.SceneManager(engine=oracle_engine)
        sm.auto_downscale = False
        sm.add_detector(psd.ContentDetector(engine=oracle_engine))
        sm.add_detector(psd.AdaptiveDetector(engine=oracle_engine))
        assert sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0)) == n
        return len(sm.get_cut_list(show_warning=False))

    run(2000)                                   # warm: libraries, scratch buffers
    base = rss()
    cuts = run(40000)
    grown = rss() - base
    assert cuts >= 100
    assert grown < 24 << 20, f"RSS grew by {grown >> 20} MiB over 40 000 frames of 7 KB"      # (all frames together: 276 MB)


@pytest.mark.parametrize("fps", [25.0, 29.97, 23.976, 60.0])
def test_threshold_cuts_do_not_depend_on_what_backs_the_positions(oracle_engine, fps):
    """The reference's cross-backend regression (``tests/release/test_backends.py:102-137``): a fade's cut frame is computed on frame
    numbers (``threshold_detector.py:146-157``), so a stream whose positions are frame numbers, one whose positions are
    sub-microsecond presentation timestamps (PyAV) and one with millisecond-truncated ones (OpenCV's CAP_PROP_POS_MSEC) give the
    same cuts -- fades of every length, so that midpoints land on .5 boundaries."""
    from fractions import Fraction

    n = 400
    level = np.full(n, 120, np.int64)
    start = 20
    for length in range(3, 22):                 # fade-outs / -ins of growing length
        level[start:start + length] = 2
        start += length + 9
    frames = np.broadcast_to(level[:, None, None, None], (n, 8, 12, 3)).astype(np.uint8)
    rate = psd.FrameTimecode(0, fps).frame_rate
    micro = [round(Fraction(i) / rate * 1_000_000) for i in range(n)]
    milli = [int(Fraction(i) / rate * 1000) for i in range(n)]          # truncated, like CAP_PROP_POS_MSEC

    def cuts(video, **kw):
        sm = psd.SceneManager(engine=oracle_engine)
        sm.add_detector(psd.ThresholdDetector(min_scene_len=2, engine=oracle_engine, **kw))
        sm.detect_scenes(video)
        return [c.frame_num for c in sm.get_cut_list(show_warning=False)]

    for kw in ({}, {"fade_bias": 0.31}, {"fade_bias": -1.0}, {"fade_bias": 1.0}):
        by_frames = cuts(psd.ArrayVideoStream(frames, fps), **kw)
        assert len(by_frames) >= 15
        assert cuts(psd.ArrayVideoStream(frames, fps, pts=micro, time_base=Fraction(1, 1_000_000)), **kw) == by_frames
        assert cuts(psd.ArrayVideoStream(frames, fps, pts=milli, time_base=Fraction(1, 1000)), **kw) == by_frames


def test_managers_on_parallel_threads_equal_serial_runs(golden, oracle_engine):
    """The reference's fan-out parity test (``tests/test_fan_out.py:157-212``: detectors in parallel threads give what serial runs
    give; detectors need not be thread-safe but instances must not share state, ``benchmark/sweep.py:160-180``): four managers, each
    with its own detectors and stream, on four threads at once."""
    import threading

    jobs = [("scenes_a", psd.ContentDetector, {}), ("fades_b", psd.ThresholdDetector, {"add_final_scene": True}),
            ("ragged_c", psd.AdaptiveDetector, {"window_width": 1}), ("scenes_a", psd.HistogramDetector, {})]

    def run(clip, cls, kw):
        stats = psd.StatsManager()
        sm = psd.SceneManager(stats, engine=oracle_engine, batch_frames=7)
        det = cls(engine=oracle_engine, **kw)
        sm.add_detector(det)
        frames = golden_clip(golden, clip)
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
        key = det.get_metrics()[0]
        return ([c.frame_num for c in sm.get_cut_list(show_warning=False)],
                [stats.get_metrics(i, [key])[0] if stats.metrics_exist(i, [key]) else None for i in range(len(frames))])

    serial = [run(*job) for job in jobs]
    for _ in range(3):
        parallel, errors = [None] * len(jobs), []

        def work(i):
            try:
                parallel[i] = run(*jobs[i])
            except BaseException as ex:  # noqa: BLE001
                errors.append(ex)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert parallel == serial
    assert any(cuts for cuts, _ in serial)
