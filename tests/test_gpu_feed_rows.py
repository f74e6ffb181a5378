"""GPU: frames of which only the rows a downscale reads were uploaded (``psd_upload_rows``, ABI 4) give the records,
thumbnails and small frames of fully uploaded ones -- through the C-ABI, against the oracle on the small shapes and
against the full upload at 1080p / 4K (reference scene_manager.py:666-678: every consumer sees the resized frame)."""
import ctypes

import numpy as np
import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import _native
from pyscenedetect_amd import engine as E
from tests.conftest import golden_clip

pytestmark = pytest.mark.gpu

NEAREST, LINEAR, CUBIC = 0, 1, 2


def _poisoned(engine, nbytes):
    buf = engine.alloc(nbytes)
    buf.upload(np.full(nbytes, 0xA5, np.uint8))
    return buf


def _analyze(engine, buf, n, h, w, factor, interp, flags, hash_sizes=()):
    return engine.analyze_device(buf.ptr, n, h, w, h * w * 3, flags=flags, downscale=factor, interpolation=interp,
                                 hash_sizes=hash_sizes, want_frames=True)


_ROW_SHAPES = [(1080, 1920, 7.5, 6), (1080, 1920, 7.0, 2), (2160, 3840, 15.0, 3), (270, 480, 3.0, 9), (97, 131, 3.0, 5), (720, 1280, 5.0, 4),
               (270, 480, 6.0, 9)]


# (CUBIC: four taps per destination row -- below a factor of 6 nearly every source row carries one and whole frames travel)
@pytest.mark.parametrize("h,w,factor,n,interp", [s + (i,) for i in (LINEAR, NEAREST, CUBIC) for s in _ROW_SHAPES if i != CUBIC or s[2] >= 6])
def test_tap_rows_only_equal_the_full_upload(hip_engine, oracle_engine, h, w, factor, n, interp):
    rng = np.random.default_rng(h + w + interp)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[n // 2:, : h // 3] //= 4
    dw, dh = max(1, round(w / factor)), max(1, round(h / factor))
    rows = hip_engine.downscale_source_rows(h, w, dh, dw, interp)
    assert len(rows) <= {LINEAR: 2, NEAREST: 1, CUBIC: 4}[interp] * dh < 0.7 * h
    stride = h * w * 3
    sparse, full = _poisoned(hip_engine, n * stride), hip_engine.alloc(n * stride)
    for t in range(n):
        sparse.upload_rows(frames[t], t * stride, rows)
    full.upload(frames.reshape(-1))
    # the rows in between really were left alone
    back = sparse.download(stride).reshape(h, w * 3)
    listed = np.zeros(h, bool)
    listed[rows] = True
    assert np.array_equal(back[listed], frames[0].reshape(h, w * 3)[listed]) and np.all(back[~listed] == 0xA5)
    flags = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM | (E.SCORE_EDGES if h <= 1080 else 0)
    a = _analyze(hip_engine, sparse, n, h, w, factor, interp, flags, hash_sizes=(32,) if min(dh, dw) >= 32 else ())
    b = _analyze(hip_engine, full, n, h, w, factor, interp, flags, hash_sizes=(32,) if min(dh, dw) >= 32 else ())
    assert a["size"] == b["size"] == (dh, dw)
    assert a["records"].tobytes() == b["records"].tobytes()
    assert np.array_equal(a["frames"], b["frames"])
    for size in b["thumbs"]:
        assert np.array_equal(a["thumbs"][size], b["thumbs"][size])
    # the fused downscale + HSV kernel (no small frame in memory) reads the same rows
    fa = hip_engine.analyze_device(sparse.ptr, n, h, w, stride, flags=E.SCORE_HSV_SAD, downscale=factor, interpolation=interp)
    fb = hip_engine.analyze_device(full.ptr, n, h, w, stride, flags=E.SCORE_HSV_SAD, downscale=factor, interpolation=interp)
    assert fa["records"].tobytes() == fb["records"].tobytes()
    if h * w <= 270 * 480:      # and both are what the oracle computes from the whole frames
        want = oracle_engine.score_host(frames, flags=flags, downscale=factor, interpolation=interp)
        for key in ("sad_h", "sad_s", "sad_v", "byte_sum", "edge_xor", "hist"):
            assert np.array_equal(a["records"][key], want[key]), key
        assert np.array_equal(a["frames"], oracle_engine.downscale_host(frames, factor, interp))
    sparse.free()
    full.free()


def test_rows_of_a_strided_host_frame(hip_engine):
    """A cropped view (row pitch > row bytes) goes row by row; single rows, row pairs and a lone tail row."""
    h, w = 64, 48
    rng = np.random.default_rng(5)
    wide = rng.integers(0, 256, (h, w + 16, 3), dtype=np.uint8)
    view = wide[:, 8: 8 + w]
    rows = np.array([0, 1, 5, 6, 10, 11, 12, 30, 63], np.int32)
    buf = _poisoned(hip_engine, h * w * 3)
    _native.check(hip_engine._lib.psd_upload_rows(hip_engine._h, buf.ptr, view.ctypes.data, w * 3, view.strides[0],
                                                  rows.ctypes.data, len(rows)))
    back = buf.download().reshape(h, w * 3)
    listed = np.zeros(h, bool)
    listed[rows] = True
    assert np.array_equal(back[listed], np.ascontiguousarray(view).reshape(h, w * 3)[listed]) and np.all(back[~listed] == 0xA5)
    # the same list from a packed frame: consecutive rows travel as one piece, equally spaced groups as one strided copy
    buf2 = _poisoned(hip_engine, h * w * 3)
    buf2.upload_rows(np.ascontiguousarray(view), 0, rows)
    assert np.array_equal(buf2.download(), buf.download())
    # errors: descending rows, rows beyond the frame (checked by the Python face), a pitch below the row length
    bad = np.array([3, 2], np.int32)
    assert hip_engine._lib.psd_upload_rows(hip_engine._h, buf.ptr, view.ctypes.data, w * 3, view.strides[0], bad.ctypes.data, 2) \
        == _native.PSD_ERR_INVALID
    assert hip_engine._lib.psd_upload_rows(hip_engine._h, buf.ptr, view.ctypes.data, w * 3, w * 3 - 1, rows.ctypes.data, 2) \
        == _native.PSD_ERR_INVALID
    with pytest.raises(ValueError):
        buf.upload_rows(np.ascontiguousarray(view), 0, np.array([64], np.int32))
    buf.free()
    buf2.free()


@pytest.mark.parametrize("h,w,n", [(1080, 1920, 40), (97, 131, 7), (64, 48, 1), (270, 480, 33)])
def test_batched_row_upload_equals_the_per_frame_upload(hip_engine, h, w, n):
    """``psd_upload_rows_batch`` (ABI 5: rows of many separately allocated frames gathered by the engine's worker threads into
    page-locked memory, one asynchronous copy + a scatter kernel per call) leaves the device frames exactly as one
    ``psd_upload_rows`` per frame does: listed rows in place, everything else untouched.  More calls than staging segments, a
    row length that is not a multiple of 16 bytes, frames that are cropped views, a frame stride with padding."""
    rng = np.random.default_rng(h * 7 + n)
    frames = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(n)]
    factor = 7.5 if h == 1080 else 3.0
    dw, dh = max(1, round(w / factor)), max(1, round(h / factor))
    rows = hip_engine.downscale_source_rows(h, w, dh, dw, LINEAR)
    stride = (h * w * 3 + 15) & ~15
    a, b = _poisoned(hip_engine, n * stride), _poisoned(hip_engine, n * stride)
    for t in range(n):
        a.upload_rows(frames[t], t * stride, rows)
    for t0 in range(0, n, 6):                       # 6 frames per call: the ring of three segments wraps around
        b.upload_rows_batch(frames[t0:t0 + 6], t0 * stride, rows, stride)
    hip_engine.upload_fence(wait_on_host=True)
    assert np.array_equal(a.download(), b.download())
    # cropped views of wider frames (row pitch > row bytes), all in one call
    wide = [rng.integers(0, 256, (h, w + 16, 3), dtype=np.uint8) for _ in range(min(n, 5))]
    views = [x[:, 8:8 + w] for x in wide]
    c, d = _poisoned(hip_engine, len(views) * stride), _poisoned(hip_engine, len(views) * stride)
    for t, v in enumerate(views):
        c.upload_rows(np.ascontiguousarray(v), t * stride, rows)
    d.upload_rows_batch(views, 0, rows, stride)
    hip_engine.upload_fence(wait_on_host=True)
    assert np.array_equal(c.download(), d.download())
    for buf in (a, b, c, d):
        buf.free()


def test_batched_row_upload_argument_errors(hip_engine):
    h, w = 32, 16
    f = np.zeros((h, w, 3), np.uint8)
    buf = hip_engine.alloc(2 * h * w * 3)
    with pytest.raises(ValueError):
        buf.upload_rows_batch([f, f, f], 0, np.array([0, 1], np.int32), h * w * 3)          # third frame beyond the buffer
    with pytest.raises(ValueError):
        buf.upload_rows_batch([f], 0, np.array([32], np.int32), h * w * 3)                  # row beyond the frame
    with pytest.raises(ValueError):
        buf.upload_rows_batch([f, np.zeros((h, w + 1, 3), np.uint8)], 0, np.array([0], np.int32), h * w * 3)
    lib = hip_engine._lib
    ptrs = (ctypes.c_void_p * 1)(f.ctypes.data)
    bad = np.array([3, 2], np.int32)
    assert lib.psd_upload_rows_batch(hip_engine._h, buf.ptr, h * w * 3, ptrs, 1, w * 3, w * 3, bad.ctypes.data, 2) == _native.PSD_ERR_INVALID
    assert lib.psd_upload_rows_batch(hip_engine._h, buf.ptr, h * w * 3, ptrs, 1, w * 3, w * 3 - 1, bad.ctypes.data, 1) == _native.PSD_ERR_INVALID
    assert lib.psd_upload_rows_batch(hip_engine._h, buf.ptr, h * w * 3, None, 1, w * 3, w * 3, bad.ctypes.data, 1) == _native.PSD_ERR_INVALID
    buf.upload_rows_batch([], 0, np.array([0], np.int32), h * w * 3)                        # nothing to do
    buf.free()


def _run_downscaled(engine, frames, factor, interpolation, detector):
    stats = psd.StatsManager()
    sm = psd.SceneManager(stats, engine=engine, batch_frames=16)
    sm.auto_downscale = False
    sm.downscale = factor
    sm.interpolation = interpolation
    sm.add_detector(detector)
    shown = []
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), callback=lambda img, pos: shown.append((pos.frame_num, int(img.sum()))))
    metrics = [[stats.get_metrics(i, [k])[0] if stats.metrics_exist(i, [k]) else None for k in detector.get_metrics()]
               for i in range(len(frames))]
    return [c.frame_num for c in sm.get_cut_list(show_warning=False)], metrics, shown


@pytest.mark.parametrize("interpolation", ["LINEAR", "NEAREST"])
def test_scene_manager_feeds_tap_rows_and_decides_like_the_oracle(golden, hip_engine, oracle_engine, interpolation):
    """SceneManager over the device feeder (tap rows only: 36 of 72 rows at factor 4) == the same manager over the oracle
    engine (whole frames, CPU): cuts, every frame metric, the frames handed to the callback."""
    from pyscenedetect_amd import scene_manager as smod

    frames = golden_clip(golden, "scenes_a")
    mode = psd.Interpolation[interpolation]
    seen = []
    original = smod._DeviceFeeder.put

    def spy(self, slot, index, frame):
        seen.append(0 if slot["rows"] is None else len(slot["rows"]))
        return original(self, slot, index, frame)

    smod._DeviceFeeder.put = spy
    try:
        got = _run_downscaled(hip_engine, frames, 4, mode, psd.ContentDetector(engine=hip_engine))
        got_a = _run_downscaled(hip_engine, frames, 4, mode, psd.AdaptiveDetector(engine=hip_engine))
    finally:
        smod._DeviceFeeder.put = original
    assert seen == [36 if interpolation == "LINEAR" else 18] * (2 * len(frames))
    want = _run_downscaled(oracle_engine, frames, 4, mode, psd.ContentDetector(engine=oracle_engine))
    want_a = _run_downscaled(oracle_engine, frames, 4, mode, psd.AdaptiveDetector(engine=oracle_engine))
    assert got == want and len(got[0]) > 0
    assert got_a == want_a


def test_reference_runs_on_larger_frames_through_the_row_feeder(golden, hip_engine):
    """golden["downscale_rows"]: the UNMODIFIED reference over whole 960 x 540 frames (auto downscale to 256 x 144, LINEAR,
    NEAREST, LANCZOS4 and CUBIC, six detector configurations) against the HIP path whose feeder uploaded 288 (144) of the 540 rows of every frame:
    same cuts, same per-frame metrics."""
    from pyscenedetect_amd import scene_manager as smod
    from tests._helpers import assert_same_run, run_config
    from tests.test_host_golden import big_clip

    frames = big_clip(golden)
    seen = []
    original = smod._DeviceFeeder.put

    def spy(self, slot, index, frame):
        seen.append(0 if slot["rows"] is None else len(slot["rows"]))
        return original(self, slot, index, frame)

    smod._DeviceFeeder.put = spy
    try:
        for mode, runs in golden["downscale_rows"]["results"].items():
            for name, want in runs.items():
                cls_name, kwargs, with_stats = golden["configs"][name]
                del seen[:]
                got = run_config(frames, cls_name, kwargs, with_stats, hip_engine, auto_downscale=True, interpolation=mode)
                assert_same_run(got, want, f"big_e/{mode}/{name}")
                # (LANCZOS4 / CUBIC: eight / four taps per destination row over 3.75 source rows -- every row carries taps, whole frames travel)
                assert seen == [{"LINEAR": 288, "NEAREST": 144, "LANCZOS4": 0, "CUBIC": 0}[mode]] * len(frames), (mode, name, seen[:3])
    finally:
        smod._DeviceFeeder.put = original


def test_default_pipeline_at_1080p_takes_the_row_path(hip_engine):
    """auto_downscale at 1080p: factor 7.5 -> 256 x 144, 288 of 1080 rows in two strided copies per frame; same cuts as with
    whole-frame uploads."""
    from pyscenedetect_amd import scene_manager as smod
    from pyscenedetect_amd.synth import make_clip

    clip, true_cuts = make_clip(11, 48, 1080, 1920, shot_len=(10, 14))
    rows = hip_engine.downscale_source_rows(1080, 1920, 144, 256, LINEAR)
    assert len(rows) == 288 and hip_engine.upload_rows_plan(rows).tolist() == [[3, 2, 15, 72], [10, 2, 15, 72]]
    seen = []
    original = smod._DeviceFeeder.put

    def spy(self, slot, index, frame):
        seen.append(0 if slot["rows"] is None else len(slot["rows"]))
        return original(self, slot, index, frame)

    def run():
        sm = psd.SceneManager(engine=hip_engine, batch_frames=16)
        sm.add_detector(psd.ContentDetector(min_scene_len=4, engine=hip_engine))
        sm.detect_scenes(psd.ArrayVideoStream(clip, 25.0))
        return [c.frame_num for c in sm.get_cut_list(show_warning=False)]

    smod._DeviceFeeder.put = spy
    keep = E.ScoringEngine.ROWS_ONLY_BELOW
    try:
        got = run()
        E.ScoringEngine.ROWS_ONLY_BELOW = 0.0      # whole frames
        full = run()
    finally:
        smod._DeviceFeeder.put = original
        E.ScoringEngine.ROWS_ONLY_BELOW = keep
    assert seen == [288] * 48 + [0] * 48
    assert got == full and len(got) >= 2 and set(got) <= set(true_cuts)


@pytest.mark.parametrize("interp", [LINEAR, NEAREST])
def test_host_frame_entry_points_upload_tap_rows(hip_engine, oracle_engine, interp):
    """score_host(downscale=...) (a stacked array: the chunk travels as one tall frame, two strided copies at 1080p) and
    score_frames / analyze_frames (separately allocated frames) behind a downscale == the oracle on whole frames."""
    rng = np.random.default_rng(21 + interp)
    for h, w, factor, n in ((1080, 1920, 7.5, 5), (270, 480, 4.0, 7), (99, 160, 4.0, 6)):
        frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        prev = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert hip_engine.tap_rows(h, w, factor, interp) is not None
        flags = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
        want = oracle_engine.score_host(frames, prev=prev, flags=flags, downscale=factor, interpolation=interp)
        got = hip_engine.score_host(frames, prev=prev, flags=flags, downscale=factor, interpolation=interp)
        got2 = hip_engine.score_frames([f.copy() for f in frames], prev=prev.copy(), flags=flags, downscale=factor, interpolation=interp)
        for key in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"):
            assert np.array_equal(got[key], want[key]), (h, key)
            assert np.array_equal(got2[key], want[key]), (h, key)
    rows = hip_engine.tap_rows(1080, 1920, 7.5, LINEAR)
    tall = (rows[None, :] + (np.arange(5, dtype=np.int32) * 1080)[:, None]).reshape(-1)
    assert len(hip_engine.upload_rows_plan(tall)) == 2


def test_table_cache_across_shapes_modes_and_engines(hip_engine):
    """NEAREST / AREA / hash-thumbnail tables are built once per (mode, shapes) and kept by the engine: interleaved shapes and
    a second engine give what a fresh call gives."""
    import cv2  # the oracle shim

    rng = np.random.default_rng(9)
    cases = [((90, 120), (30, 40)), ((97, 131), (41, 77)), ((90, 120), (45, 60))]
    srcs = {c: rng.integers(0, 256, (2, c[0][0], c[0][1], 3), dtype=np.uint8) for c in cases}
    other = E.ScoringEngine(0)
    try:
        for rnd in range(3):
            for eng in (hip_engine, other):
                for c in cases:
                    (sh, sw), (dh, dw) = c
                    a = eng.alloc(srcs[c].nbytes)
                    a.upload(srcs[c].reshape(-1))
                    b = eng.alloc(2 * dh * dw * 3)
                    for inter in (cv2.INTER_NEAREST, cv2.INTER_AREA):
                        eng.resize_device(a.ptr, 2, sh, sw, b.ptr, dh, dw, interpolation=inter)
                        got = b.download().reshape(2, dh, dw, 3)
                        for i in range(2):
                            assert np.array_equal(got[i], cv2.resize(srcs[c][i], (dw, dh), interpolation=inter)), (rnd, c, inter)
                    for size in (16, 24):
                        thumbs = eng.hash_thumbs_device(a.ptr, 2, sh, sw, size)
                        for i in range(2):
                            gray = cv2.cvtColor(srcs[c][i], cv2.COLOR_BGR2GRAY)
                            assert np.array_equal(thumbs[i], cv2.resize(gray, (size, size), interpolation=cv2.INTER_AREA)), (rnd, c, size)
                    a.free()
                    b.free()
    finally:
        other.close()


@pytest.mark.gpu
def test_cpus_near_the_device(hip_engine):
    """``psd_cpus_near_device`` (ABI 6): ascending CPUs inside this thread's affinity mask, a proper subset of it or nothing;
    the count comes back whole when the caller's array is short; argument errors are reported."""
    import os

    mine = os.sched_getaffinity(0)
    cpus = hip_engine.cpus_near_gpu()
    assert cpus == sorted(set(cpus)) and set(cpus) <= mine
    assert len(cpus) == 0 or len(cpus) < len(mine)
    lib, h = hip_engine._lib, hip_engine._h
    n = ctypes.c_int(-1)
    short = (ctypes.c_int * 2)()
    assert lib.psd_cpus_near_device(h, short, 2, ctypes.byref(n)) == 0 and n.value == len(cpus)
    assert list(short)[:min(2, len(cpus))] == cpus[:2]
    assert lib.psd_cpus_near_device(h, None, 0, ctypes.byref(n)) == 0 and n.value == len(cpus)
    assert lib.psd_cpus_near_device(h, None, 4, ctypes.byref(n)) == _native.PSD_ERR_INVALID
    assert lib.psd_cpus_near_device(h, short, 2, None) < 0
    assert os.sched_getaffinity(0) == mine           # asking moves nobody


@pytest.mark.gpu
@pytest.mark.parametrize("auto_downscale", [False, True])
def test_one_manager_on_two_videos_carries_the_last_frame(golden, hip_engine, oracle_engine, auto_downscale):
    """A manager run on a second video without ``clear()`` scores its first frame against the first video's last one, like the
    reference's detectors (tests/test_host_fuzz_vs_reference.py): through the device feeder that frame is re-uploaded as the first
    batch's predecessor (``_DeviceFeeder.seed_halo``); with ``clear()`` in between nothing carries over."""
    a = golden_clip(golden, "wide_d" if auto_downscale else "scenes_a")
    b = a[: len(a) // 2][::-1].copy()          # starts in another shot than `a` ends in

    def run(engine, clear_between, with_stats):
        stats = psd.StatsManager() if with_stats else None
        sm = psd.SceneManager(stats, engine=engine, batch_frames=16)
        sm.auto_downscale = auto_downscale

        def detectors():
            return [psd.ContentDetector(threshold=12.0, min_scene_len=0, engine=engine),
                    psd.AdaptiveDetector(min_content_val=5.0, min_scene_len=2, engine=engine), psd.HistogramDetector(engine=engine)]

        dets = detectors()
        for d in dets:
            sm.add_detector(d)
        sm.detect_scenes(psd.ArrayVideoStream(a, 25.0))
        first = [c.frame_num for c in sm.get_cut_list(show_warning=False)]
        if clear_between:
            sm.clear()
            # "same": the SAME detector objects again -- in the reference they still hold what they derived from the last frame they
            # saw (clear() only empties the manager's list, scene_manager.py:358-375), so the first video's last frame carries over;
            # "fresh": new detectors, nothing carries over
            for d in (dets if clear_between == "same" else detectors()):
                sm.add_detector(d)
        sm.detect_scenes(psd.ArrayVideoStream(b, 25.0))
        return first, [c.frame_num for c in sm.get_cut_list(show_warning=False)]

    results = {}
    for clear_between in (False, "same", "fresh"):
        for with_stats in (False, True):
            got, want = run(hip_engine, clear_between, with_stats), run(oracle_engine, clear_between, with_stats)
            assert got == want, (clear_between, with_stats)
            results[clear_between, with_stats] = got
    # the carried frame shows: the second video's frame 0 is a cut when its predecessor is another shot's frame -- without clear(), and
    # with clear() when the same detectors come back (the reference: tools/fuzz_host_vs_reference.py, mode "mixed"); not with fresh ones
    assert 0 in results[False, False][1] and 0 in results["same", False][1] and 0 not in results["fresh", False][1]


class _MeanJump(psd.SceneDetector):
    """A detector of the caller's own on the plug-in API only (no device path): it must be handed the downscaled frame."""

    def __init__(self):
        super().__init__()
        self.seen, self.last = [], None

    def process_frame(self, timecode, frame_img):
        self.seen.append((timecode.frame_num, frame_img.shape, int(np.asarray(frame_img, dtype=np.int64).sum())))
        mean = float(frame_img.mean())
        cut = self.last is not None and abs(mean - self.last) >= 20.0
        self.last = mean
        return [timecode] if cut else []

    def get_metrics(self):
        return []


@pytest.mark.parametrize("interpolation", ["LINEAR", "AREA"])
def test_a_manager_of_plug_in_detectors_only_hands_them_downscaled_frames(hip_engine, oracle_engine, interpolation):
    """reference scene_manager.py:666-678: every consumer sees the resized frame -- also when no built-in detector is registered
    and nothing is scored (analyze_device(flags=0, want_frames=True) behind the row feeder)."""
    frames = np.random.default_rng(8).integers(0, 256, (40, 270, 480, 3), dtype=np.uint8)
    frames[17:] //= 3
    runs = []
    for engine in (hip_engine, oracle_engine):
        det = _MeanJump()
        sm = psd.SceneManager(engine=engine, batch_frames=16)
        sm.auto_downscale = False
        sm.downscale = 3
        sm.interpolation = psd.Interpolation[interpolation]
        sm.add_detector(det)
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
        runs.append((det.seen, [c.frame_num for c in sm.get_cut_list(show_warning=False)]))
    assert runs[0] == runs[1]
    assert runs[0][0][0][1] == (90, 160, 3) and runs[0][1] == [17]
    assert np.array_equal(hip_engine.downscale_host(frames[:5], 3, psd.Interpolation[interpolation].value),
                          oracle_engine.downscale_host(frames[:5], 3, psd.Interpolation[interpolation].value))


def test_callback_of_a_later_call_gets_the_downscaled_frame_an_earlier_call_buffered(hip_engine, oracle_engine):
    """Detection in pieces: call 1 (no callback) buffers frame 20 at full size, call 2 reports the cut at 20 two frames late and
    hands the buffered frame to its callback -- downscaled then (ScoringEngine.downscale_host), like the reference's buffer holds it."""
    frames = np.random.default_rng(7).integers(118, 123, (50, 111, 480, 3), dtype=np.uint8)
    frames[20:] += 90
    runs = []
    for engine in (hip_engine, oracle_engine):
        sm = psd.SceneManager(engine=engine, batch_frames=7)
        sm.auto_downscale = False
        sm.downscale = 3
        sm.add_detector(psd.AdaptiveDetector(engine=engine, weights=psd.ContentDetector.Components(1.0, 1.0, 2.0, 0.0)))
        video = psd.ArrayVideoStream(frames, 24.0)
        shown = []
        sm.detect_scenes(video, duration=21)
        sm.detect_scenes(video, callback=lambda img, pos: shown.append((pos.frame_num, img.shape, int(img.astype(np.int64).sum()))))
        runs.append((shown, [c.frame_num for c in sm.get_cut_list(show_warning=False)]))
    assert runs[0] == runs[1]
    assert runs[0][1] == [20] and [s[:2] for s in runs[0][0]] == [(20, (37, 160, 3))]


def test_a_slice_of_the_engines_fuzz_with_the_wide_ingredients(hip_engine, oracle_engine):
    """``tools/fuzz_host_vs_reference.py --engines --wide`` (the mirror over the HIP engine against the mirror over the oracle engine: device
    feeder, tap rows, crop / downscale modes, batch sizes, the carried frame, the resident per-frame path) with the ingredients added after
    the last GPU run of round 5 -- detection from a seek position and in pieces, a plug-in detector alone or beside the others, frame
    layouts, detect(), object min_scene_len -- and every comparison on (exception texts, metric types, warnings, logs, public state).  The
    original ingredients ran on hardware (1.6 k cases, `profiles/r05_y_*`); on CPU the same cases pass over the simulated device engine.
    (Round 5 kept this slice behind an opt-in switch and its two neighbours above behind xfail marks because they had been written
    after that round's GPU minutes were spent; the driver's run showed them passing, and all three are unconditional now.)"""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "tools")]
    import fuzz_host_vs_reference as F

    F.WIDE = True
    try:
        for case in range(60):
            rng = np.random.default_rng([20250926, case])
            frames, fps, dets, cfg = F.draw_case(rng)
            cfg["batch_frames"] = int(rng.choice([1, 7, 64]))
            a = F.outcome(lambda: F.run_side("mirror", frames, fps, dets, cfg, oracle_engine))
            b = F.outcome(lambda: F.run_side("mirror", frames, fps, dets, cfg, hip_engine))
            assert F.differ(a, b, cfg) is None, (case, F.differ(a, b, cfg), list(frames.shape), dets, cfg)
    finally:
        F.WIDE = False
