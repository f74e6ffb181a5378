"""CPU: the host mirror (detectors + SceneManager + C epilogues) fed by the ORACLE's integer
records must reproduce the reference's own runs (tests/golden/reference_runs.json, produced by
oracle/gen_golden.py from the unmodified reference): identical cut lists, scene lists and
bit-identical float metrics."""
import math

import numpy as np
import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import FlashFilter, FrameTimecode, epilogue
from tests._helpers import assert_same_run, run_config
from tests.conftest import golden_clip

SMALL_CLIPS = ["scenes_a", "fades_b", "ragged_c", "uniform_u"]


def _configs(golden, clip):
    res = golden["clips"][clip]["results"]
    return [(name, golden["configs"][name]) for name in res]


@pytest.mark.parametrize("clip", SMALL_CLIPS)
def test_scene_manager_matches_reference(golden, oracle_engine, clip):
    frames = golden_clip(golden, clip)
    for name, (cls_name, kwargs, with_stats) in _configs(golden, clip):
        got = run_config(frames, cls_name, kwargs, with_stats, oracle_engine)
        assert_same_run(got, golden["clips"][clip]["results"][name], f"{clip}/{name}")


def test_batch_size_does_not_matter(golden, oracle_engine):
    frames = golden_clip(golden, "ragged_c")
    for name in ("content_stats", "adaptive_default", "hist_default", "threshold_final"):
        cls_name, kwargs, with_stats = golden["configs"][name]
        for bf in (1, 7, 1000):
            got = run_config(frames, cls_name, kwargs, with_stats, oracle_engine, batch_frames=bf)
            assert_same_run(got, golden["clips"]["ragged_c"]["results"][name], f"ragged_c/{name}/batch{bf}")


def test_process_frame_api_matches_reference(golden, oracle_engine):
    """The one-frame-at-a-time plug-in API (SceneDetector.process_frame) gives the same cuts."""
    frames = golden_clip(golden, "fades_b")
    for name in ("content_default", "adaptive_default", "hist_default", "threshold_final", "hash_default", "hash_16_lp2"):
        cls_name, kwargs, _ = golden["configs"][name]
        from tests._helpers import build_detector

        det = build_detector(cls_name, kwargs, oracle_engine)
        cuts = []
        for i, f in enumerate(frames):
            cuts += det.process_frame(FrameTimecode(i, 25.0), f)
        cuts += det.post_process(FrameTimecode(len(frames) - 1, 25.0))
        assert sorted({c.frame_num for c in cuts}) == golden["clips"]["fades_b"]["results"][name]["cuts"], name


def test_all_detectors_share_one_pass(golden, oracle_engine):
    """All five detectors on one SceneManager (one upload, one fused scoring pass + one thumbnail pass) == five separate runs."""
    frames = golden_clip(golden, "scenes_a")
    names = ["content_default", "adaptive_default", "hist_default", "threshold_default", "hash_default"]
    want = set()
    for n in names:
        want |= set(golden["clips"]["scenes_a"]["results"][n]["cuts"])
    from tests._helpers import build_detector

    sm = psd.SceneManager(engine=oracle_engine)
    sm.auto_downscale = False
    for n in names:
        cls_name, kwargs, _ = golden["configs"][n]
        sm.add_detector(build_detector(cls_name, kwargs, oracle_engine))
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == sorted(want)


# ---- pure state machines: known-answer traces generated from the reference ----------------------

def test_flash_filter_kats(golden):
    for key, kat in golden["kats"].items():
        if not key.startswith("flash_"):
            continue
        _, mode, length = key.split("_", 2)
        length = eval(length)  # repr of int/float/str written by gen_golden.py
        f = FlashFilter(FlashFilter.Mode[mode], length)
        emitted = []
        above = set(kat["above"])
        for i in range(kat["n"]):
            emitted += [[i, c.frame_num] for c in f.filter(FrameTimecode(i, 25.0), i in above)]
        assert emitted == kat["emitted"], key
        # (recorded after the run: an int length has been converted to seconds by then, detector.py:176)
        assert f.max_behind == kat["max_behind"]
        # native epilogue: same cuts (emission time is not part of its output)
        cv = np.array([30.0 if i in above else 0.0 for i in range(kat["n"])])
        cuts = epilogue.content_cuts(cv, 25.0, threshold=27.0, min_scene_len=length, filter_mode=0 if mode == "MERGE" else 1)
        assert cuts == [c for _, c in kat["emitted"]], key


def test_injected_score_kats(golden, oracle_engine):
    kat = golden["kats"]["content_injected"]
    scores = np.full(kat["n"], kat["default"])
    for k, v in kat["scores"].items():
        scores[int(k)] = v
    assert epilogue.content_cuts(scores, 25.0) == kat["cuts"]

    class Injected(psd.ContentDetector):
        def _score_from_record(self, timecode, record, height, width):
            return float(scores[timecode.frame_num])

    d = Injected(engine=oracle_engine)
    got = []
    for i in range(kat["n"]):
        got += [c.frame_num for c in d.process_record(FrameTimecode(i, 25.0), None, 4, 4)]
    assert got == kat["cuts"]

    kat = golden["kats"]["adaptive_injected"]
    scores = np.full(kat["n"], kat["default"])
    for k, v in kat["scores"].items():
        scores[int(k)] = v
    cuts, _ = epilogue.adaptive_cuts(scores, 25.0)
    assert cuts == kat["cuts"]

    class InjectedA(psd.AdaptiveDetector):
        def _score_from_record(self, timecode, record, height, width):
            return float(scores[timecode.frame_num])

    d = InjectedA(engine=oracle_engine)
    got = []
    for i in range(kat["n"]):
        got += [c.frame_num for c in d.process_record(FrameTimecode(i, 25.0), None, 4, 4)]
    assert got == kat["cuts"]


# ---- native (C++) epilogues vs the reference runs ------------------------------------------------

def _nan_to_none(a):
    return [None if math.isnan(x) else float(x) for x in a]


@pytest.mark.parametrize("clip", SMALL_CLIPS)
def test_native_epilogues_match_reference(golden, oracle_engine, clip):
    frames = golden_clip(golden, clip)
    h, w = frames.shape[1:3]
    res = golden["clips"][clip]["results"]
    recs_e = oracle_engine.score_host(frames, flags=15)
    for name in res:
        cls_name, kwargs, _ = golden["configs"][name]
        want = res[name]
        kw = dict(kwargs)
        if cls_name in ("ContentDetector", "AdaptiveDetector"):
            weights = kw.get("weights", [1.0, 1.0, 1.0, 0.0])
            if kw.get("luma_only"):
                weights = [0.0, 0.0, 1.0, 0.0]
            recs = recs_e
            if kw.get("kernel_size"):
                recs = oracle_engine.score_host(frames, flags=15, edge_kernel=kw["kernel_size"])
            sc = epilogue.content_scores(recs, h, w, weights)
            if cls_name == "ContentDetector":
                cuts = epilogue.content_cuts(sc["content_val"], 25.0, kw.get("threshold", 27.0), kw.get("min_scene_len", 15),
                                             1 if kw.get("filter_mode") == "SUPPRESS" else 0)
                if want["metrics"]:
                    for key in ("content_val", "delta_hue", "delta_sat", "delta_lum", "delta_edges"):
                        assert [None] + [float(x) for x in sc[key][1:]] == want["metrics"][key], f"{clip}/{name}/{key}"
            else:
                cuts, ratio = epilogue.adaptive_cuts(sc["content_val"], 25.0, kw.get("adaptive_threshold", 3.0),
                                                     kw.get("min_scene_len", 15), kw.get("window_width", 2),
                                                     kw.get("min_content_val", 15.0))
                key = [k for k in want["metrics"] if k.startswith("adaptive_ratio")][0]
                assert _nan_to_none(ratio) == want["metrics"][key], f"{clip}/{name}"
        elif cls_name == "HistogramDetector":
            cuts, diff = epilogue.hist_cuts(recs_e, 25.0, kw.get("threshold", 0.2), kw.get("bins", 128), kw.get("min_scene_len", 15))
            key = list(want["metrics"])[0]
            assert _nan_to_none(diff) == want["metrics"][key], f"{clip}/{name}"
        elif cls_name == "HashDetector":
            thumbs = oracle_engine.hash_thumbs_host(frames, kw.get("size", 8) * kw.get("lowpass", 2))
            bits = epilogue.hash_bits(thumbs, kw.get("size", 8))
            cuts, dist = epilogue.hash_cuts(bits, 25.0, kw.get("threshold", 0.35), kw.get("min_scene_len", 15))
            if want["metrics"]:
                assert _nan_to_none(dist) == list(want["metrics"].values())[0], f"{clip}/{name}"
        else:
            cuts, avg = epilogue.threshold_cuts(recs_e, h, w, 25.0, kw.get("threshold", 12), kw.get("min_scene_len", 15),
                                                kw.get("fade_bias", 0.0), kw.get("add_final_scene", False),
                                                1 if kw.get("method") == "CEILING" else 0)
            assert [float(x) for x in avg] == want["metrics"]["average_rgb"], f"{clip}/{name}"
        assert sorted(set(cuts)) == want["cuts"], f"{clip}/{name}: {cuts} vs {want['cuts']}"


def test_auto_downscale_matches_reference(golden, oracle_engine):
    """> 256 px wide: the SceneManager asks the engine to downscale to the reference's target size."""
    frames = golden_clip(golden, "wide_d")
    for name in golden["clips"]["wide_d"]["results"]:
        cls_name, kwargs, with_stats = golden["configs"][name]
        got = run_config(frames, cls_name, kwargs, with_stats, oracle_engine, auto_downscale=True)
        assert_same_run(got, golden["clips"]["wide_d"]["results"][name], f"wide_d/{name}")


def test_downscale_interpolation_modes_match_reference(golden, oracle_engine):
    """SceneManager.interpolation = NEAREST / AREA / LANCZOS4 / CUBIC (reference scene_manager.py:265-272): the downscale in front of
    the detectors uses the requested cv2.resize filter (CUBIC: the default form of PSD_CUBIC_FORM, which is what the goldens were made with)."""
    frames = golden_clip(golden, "wide_d")
    for mode, runs in golden["interp"].items():
        for name, want in runs.items():
            cls_name, kwargs, with_stats = golden["configs"][name]
            got = run_config(frames, cls_name, kwargs, with_stats, oracle_engine, auto_downscale=True, interpolation=mode)
            assert_same_run(got, want, f"wide_d/{mode}/{name}")
    sm = psd.SceneManager(engine=oracle_engine)
    for mode in psd.Interpolation:                             # (all five filters of the reference's enum are accepted)
        sm.interpolation = mode
        assert sm.interpolation is mode
    with pytest.raises(ValueError):
        sm.interpolation = 5


def big_clip(golden):
    """The 960 x 540 clip of golden["downscale_rows"], regenerated from its seed."""
    from pyscenedetect_amd.synth import make_clip

    c = golden["downscale_rows"]["clip"]
    kw = dict(c["kwargs"])
    kw["shot_len"] = tuple(kw["shot_len"])
    frames, _ = make_clip(c["seed"], c["n"], c["h"], c["w"], **kw)
    assert int(frames.sum()) == c["sum_all"], "synthetic clip differs from the one the golden run used"
    return frames


def test_default_downscale_of_larger_frames_matches_reference(golden, oracle_engine):
    """960 x 540 -> 256 x 144 (factor 3.75: two of every 3.75 source rows carry bilinear taps): the reference's runs over whole
    frames, LINEAR and NEAREST, six detector configurations."""
    frames = big_clip(golden)
    for mode, runs in golden["downscale_rows"]["results"].items():
        for name, want in runs.items():
            cls_name, kwargs, with_stats = golden["configs"][name]
            got = run_config(frames, cls_name, kwargs, with_stats, oracle_engine, auto_downscale=True, interpolation=mode)
            assert_same_run(got, want, f"big_e/{mode}/{name}")
    assert golden["downscale_rows"]["results"]["LINEAR"]["content_stats"]["cuts"]


# ---- plug-in surface: constructor validation, metric keys, look-behind (oracle/gen_api_golden.py) -------------------

def _api_cases():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "api_cases.json")) as fh:
        return json.load(fh)


def _outcome(fn):
    try:
        return {"ok": fn()}
    except Exception as ex:  # noqa: BLE001
        return {"raises": type(ex).__name__}


def test_constructors_and_plugin_properties_match_reference(oracle_engine):
    from tests._helpers import build_detector

    for c in _api_cases()["ctor"]:
        def describe():
            det = build_detector(c["cls"], c["kwargs"], oracle_engine)
            return {"metrics": list(det.get_metrics()), "event_buffer_length": int(det.event_buffer_length),
                    "stats_manager_is_none": det.stats_manager is None}
        want = {k: c[k] for k in ("ok", "raises") if k in c}
        assert _outcome(describe) == want, f"{c['cls']}({c['kwargs']})"


def test_rejected_frames_and_manager_properties_match_reference(oracle_engine):
    cases = _api_cases()
    tc = FrameTimecode(0, 25.0)
    for name, c in cases["bad_frames"].items():
        frame = np.zeros(c["shape"], np.dtype(c["dtype"]))
        got = _outcome(lambda: [x.frame_num for x in getattr(psd, c["cls"])(engine=oracle_engine).process_frame(tc, frame)])
        assert got == {k: c[k] for k in ("ok", "raises") if k in c}, name
    for c in cases["downscale_factor"]:
        got = _outcome(lambda: psd.compute_downscale_factor(c["width"], c["effective"]))
        assert got == {k: c[k] for k in ("ok", "raises") if k in c}, c
    sm = psd.SceneManager(engine=oracle_engine)
    sizes = []
    for cls, kw in (("ThresholdDetector", {}), ("ContentDetector", {}), ("AdaptiveDetector", {"window_width": 4}),
                    ("ContentDetector", {"min_scene_len": 40})):
        sm.add_detector(getattr(psd, cls)(engine=oracle_engine, **kw))
        sizes.append(sm._frame_buffer_size)
    assert {"ok": sizes} == cases["frame_buffer_sizes"]


def test_scene_manager_surface_matches_reference(golden, oracle_engine):
    """Property setters (downscale, crop), argument checks of detect_scenes, lists before a run, clear(), crop at and beyond
    the frame border, duration / end_time: every step's value or exception type AND the warnings logged on the way, as the
    reference's SceneManager produced them (oracle/manager_cases.py, scene_manager.py:278-335, 358-372, 446-530)."""
    from oracle.manager_cases import manager_cases

    frames = golden_clip(golden, "scenes_a")
    got = manager_cases(lambda: frames,
                        lambda with_stats: psd.SceneManager(psd.StatsManager() if with_stats else None, engine=oracle_engine),
                        lambda fr: psd.ArrayVideoStream(fr, 25.0), lambda: psd.ContentDetector(engine=oracle_engine))
    want = _api_cases()["manager_ops"]
    assert set(got) == set(want)
    for name in want:
        assert got[name] == want[name], f"{name}: {got[name]} != {want[name]}"


def test_variable_frame_rate_positions_match_reference(golden, oracle_engine):
    """PTS-backed positions (what the reference's PyAV backend reports for VFR video) through SceneManager and the
    detectors: cuts and scene boundaries carry the same presentation timestamps as the reference's."""
    from fractions import Fraction

    from tests._helpers import build_detector

    v = golden["vfr"]
    frames = golden_clip(golden, v["clip"])
    rng = np.random.default_rng(v["pts_seed"])
    steps = rng.choice([20, 40, 40, 40, 40, 60, 80], size=len(frames) - 1)
    pts = [0] + [int(x) for x in np.cumsum(steps)]
    for name, want in v["results"].items():
        cls_name, kwargs, with_stats = golden["configs"][name]
        sm = psd.SceneManager(psd.StatsManager() if with_stats else None, engine=oracle_engine, batch_frames=48)
        sm.auto_downscale = False
        sm.add_detector(build_detector(cls_name, kwargs, oracle_engine))
        n = sm.detect_scenes(psd.ArrayVideoStream(frames, v["fps"], pts=pts, time_base=Fraction(*v["time_base"])))
        assert n == want["frames_processed"], name
        cuts = [[c.frame_num, c.pts, c.seconds, c.get_timecode()] for c in sm.get_cut_list(show_warning=False)]
        assert cuts == want["cuts"], f"{name}: {cuts} vs {want['cuts']}"
        scenes = [[a.pts, b.pts, a.get_timecode(), b.get_timecode()] for a, b in sm.get_scene_list()]
        assert scenes == want["scenes"], name


def test_flash_filter_random_sequences_match_reference():
    """FlashFilter fed with random above/below sequences: same cuts, emitted while processing the same frames."""
    for c in _api_cases()["flash_filter"]:
        def drive():
            flt = FlashFilter(FlashFilter.Mode[c["mode"]], c["length"])
            emitted = []
            for i, a in enumerate(c["above"]):
                for cut in flt.filter(FrameTimecode(i, c["fps"]), bool(a)):
                    emitted.append([i, cut.frame_num])
            return {"emitted": emitted, "max_behind": int(flt.max_behind)}
        assert _outcome(drive) == {k: c[k] for k in ("ok", "raises") if k in c}, (c["fps"], c["length"], c["mode"])


def test_flash_filter_random_pts_sequences_match_reference():
    """The same on presentation-timestamp positions (variable frame rate): gaps are time differences."""
    from fractions import Fraction

    from pyscenedetect_amd.timecode import Timecode

    cases = _api_cases()["flash_filter_pts"]
    assert len(cases) >= 36
    for c in cases:
        def drive():
            flt = FlashFilter(FlashFilter.Mode[c["mode"]], c["length"])
            emitted = []
            for i, a in enumerate(c["above"]):
                tc = FrameTimecode(Timecode(c["pts"][i], Fraction(1, 1000)), 25.0)
                for cut in flt.filter(tc, bool(a)):
                    emitted.append([i, cut.pts])
            return {"emitted": emitted, "max_behind": int(flt.max_behind)}
        assert _outcome(drive) == {k: c[k] for k in ("ok", "raises") if k in c}, (c["length"], c["mode"])


@pytest.mark.parametrize("cls_name", ["ContentDetector", "AdaptiveDetector", "HistogramDetector", "HashDetector", "ThresholdDetector"])
def test_min_scene_len_accepts_time_values(golden, oracle_engine, cls_name):
    """int frames, float seconds and both string forms of the same duration give the same cut list
    (the reference checks this on goldeneye.mp4, tests/test_detectors.py:236-262; here 15 frames at 25 fps)."""
    frames = golden_clip(golden, "scenes_a")
    results = []
    for min_scene_len in (15, 0.6, "0.6s", "00:00:00.600"):
        got = run_config(frames, cls_name, {"min_scene_len": min_scene_len}, False, oracle_engine)
        results.append(got["cuts"])
    assert results[0] == results[1] == results[2] == results[3], results
    default_name = {"ContentDetector": "content_default", "AdaptiveDetector": "adaptive_default", "HistogramDetector": "hist_default",
                    "HashDetector": "hash_default", "ThresholdDetector": "threshold_default"}[cls_name]
    assert results[0] == golden["clips"]["scenes_a"]["results"][default_name]["cuts"]


def test_static_helpers_match_the_reference_lines(oracle_engine):
    """``HistogramDetector.calculate_histogram`` (static, histogram_detector.py:122-165: cvtColor(BGR2YUV) + split + calcHist,
    normalised and flattened or raw counts as a column) and ``_estimated_kernel_size`` (content_detector.py:39-46)."""
    import math

    import cv2  # the oracle's shim

    from pyscenedetect_amd.detectors.content_detector import estimated_kernel_size

    rng = np.random.default_rng(12)
    frame = rng.integers(0, 256, (45, 64, 3), dtype=np.uint8)
    frame[:20] //= 3
    for bins in (256, 128, 100, 16):
        y, _, _ = cv2.split(cv2.cvtColor(frame, cv2.COLOR_BGR2YUV))
        raw = cv2.calcHist([y], [0], None, [bins], [0, 256])
        got_raw = psd.HistogramDetector.calculate_histogram(frame, bins=bins, normalize=False, engine=oracle_engine)
        assert got_raw.shape == raw.shape == (bins, 1) and got_raw.dtype == raw.dtype == np.float32
        assert np.array_equal(got_raw, raw)
        want = cv2.normalize(raw.copy(), None).flatten()
        got = psd.HistogramDetector.calculate_histogram(frame, bins=bins, engine=oracle_engine)
        assert got.shape == want.shape == (bins,) and got.dtype == want.dtype
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        psd.HistogramDetector.calculate_histogram(frame.astype(np.uint16), engine=oracle_engine)
    with pytest.raises(ValueError):
        psd.HistogramDetector.calculate_histogram(frame[:, :, :2], engine=oracle_engine)
    for (w, h), k in (((1920, 1080), 13), ((3840, 2160), 19), ((256, 144), 5), ((320, 180), 5), ((1280, 720), 9), ((7680, 4320), 35)):
        size = 4 + round(math.sqrt(w * h) / 192)
        assert estimated_kernel_size(w, h) == (size + 1 if size % 2 == 0 else size) == k
