"""CPU: the host feeder uploads only the rows a downscale reads (``psd_resize_source_rows`` / ``psd_upload_rows``, ABI 4).

The row list comes from the product's own coefficient tables; here it is tied to the oracle's restatement of
``cv2.resize`` (reference scene_manager.py:666-678: every frame is resized before any detector or callback sees it):
whatever the rows NOT listed hold, the resized frame is the same.  The feeder's routing is run end to end over a
host-memory stand-in of the device engine whose un-uploaded rows are poisoned."""
import ctypes

import numpy as np
import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import _native
from pyscenedetect_amd.engine import TapRowPolicy
from pyscenedetect_amd.scene_manager import _DeviceFeeder
from tests.conftest import golden_clip

NEAREST, LINEAR, AREA = 0, 1, 3


def source_rows(h, w, dh, dw, interp):
    rows = np.empty(h, np.int32)
    n = ctypes.c_int(0)
    _native.check(_native.load().psd_resize_source_rows(h, w, dh, dw, interp, rows.ctypes.data, ctypes.addressof(n)))
    return rows[: n.value].copy()


SHAPES = [(1080, 1920, 7.0), (720, 1280, 5.0), (480, 854, 3.0), (270, 481, 1.5), (96, 128, 2.0), (97, 131, 2.0), (33, 47, 4.2)]


@pytest.mark.parametrize("interp", [NEAREST, LINEAR, AREA, 2, 4])       # (2 = CUBIC, 4 = LANCZOS4)
@pytest.mark.parametrize("h,w,factor", SHAPES)
def test_rows_not_listed_never_reach_the_resized_frame(h, w, factor, interp):
    import cv2  # the oracle's shim (tests/conftest.py puts it on the path)

    dw, dh = max(1, round(w / factor)), max(1, round(h / factor))
    rows = source_rows(h, w, dh, dw, interp)
    assert len(rows) and np.all(np.diff(rows) > 0) and rows[0] >= 0 and rows[-1] < h
    if interp == LINEAR and not (h == 2 * dh and w == 2 * dw):
        assert len(rows) <= 2 * dh
    if interp == NEAREST:
        assert len(rows) <= dh
    rng = np.random.default_rng(h * 131 + w + interp)
    frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = cv2.resize(frame, (dw, dh), interpolation=interp)
    poisoned = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    poisoned[rows] = frame[rows]
    assert np.array_equal(cv2.resize(poisoned, (dw, dh), interpolation=interp), want)
    # and the list is not padded: every listed row matters for LINEAR / NEAREST on random content (a row whose weight
    # is zero would be the exception; none at these shapes for NEAREST)
    if interp == NEAREST:
        for r in rows[:: max(1, len(rows) // 8)]:
            broken = frame.copy()
            broken[r] ^= 0xFF
            assert not np.array_equal(cv2.resize(broken, (dw, dh), interpolation=interp), want)


def test_exact_halving_and_area_need_every_row():
    assert len(source_rows(96, 128, 48, 64, LINEAR)) == 96        # OpenCV averages 2x2 boxes there
    assert len(source_rows(1080, 1920, 154, 274, AREA)) == 1080
    # INTER_AREA that enlarges runs on the bilinear passes: two rows per destination row at most
    assert len(source_rows(40, 60, 80, 120, AREA)) == 40


def test_source_rows_argument_errors():
    lib = _native.load()
    rows = np.empty(16, np.int32)
    n = ctypes.c_int(0)
    assert lib.psd_resize_source_rows(0, 16, 4, 4, LINEAR, rows.ctypes.data, ctypes.addressof(n)) == _native.PSD_ERR_INVALID
    assert lib.psd_resize_source_rows(16, 16, 4, 4, LINEAR, None, ctypes.addressof(n)) == _native.PSD_ERR_INVALID
    assert lib.psd_resize_source_rows(16, 16, 4, 4, 5, rows.ctypes.data, ctypes.addressof(n)) == _native.PSD_ERR_UNSUPPORTED  # (no such filter)
    assert "not one of cv2's filters" in _native.last_error()
    assert lib.psd_upload_rows(None, None, None, 16, 16, None, 0) == _native.PSD_ERR_INVALID


def copy_plan(rows, packed=True):
    lib = _native.load()
    rows = np.ascontiguousarray(rows, np.int32)
    n = ctypes.c_int(0)
    _native.check(lib.psd_upload_rows_plan(rows.ctypes.data, len(rows), int(packed), None, 0, ctypes.addressof(n)))
    plan = np.zeros((n.value, 4), np.int32)
    _native.check(lib.psd_upload_rows_plan(rows.ctypes.data, len(rows), int(packed), plan.ctypes.data, n.value, ctypes.addressof(n)))
    return plan


def expand(plan):
    out = []
    for first, length, step, count in plan:
        for k in range(count):
            out += list(range(first + k * step, first + k * step + length))
    return out


@pytest.mark.parametrize("h,w,factor,copies", [(1080, 1920, 7.5, 2), (1080, 1920, 7.0, 2), (2160, 3840, 15.0, 1), (720, 1280, 5.0, 1),
                                               (480, 854, 3.3359375, 3), (1080, 1920, 4.3, None)])
def test_copy_plan_covers_every_row_once(h, w, factor, copies):
    """The strided copies of ``psd_upload_rows``: every listed row exactly once, the default pipeline's 1080p -> 256 x 144
    (rows 3,4 | 10,11 | 18,19 ...: steps of 7 and 8) as two interleaved progressions of step 15."""
    dw, dh = max(1, round(w / factor)), max(1, round(h / factor))
    for interp in (NEAREST, LINEAR):
        rows = source_rows(h, w, dh, dw, interp)
        for packed in (True, False):
            plan = copy_plan(rows, packed)
            got = expand(plan)
            assert sorted(got) == list(rows) and len(got) == len(rows)
            assert np.all(plan[:, 1] >= 1) and np.all(plan[:, 3] >= 1)
            assert np.all((plan[:, 3] == 1) | (plan[:, 2] >= plan[:, 1]))           # pitch >= width of a strided copy
            if not packed:
                assert np.all(plan[:, 1] == 1)
            elif copies is not None:
                assert len(plan) == copies, (interp, plan)
    if copies is None:
        assert len(copy_plan(source_rows(h, w, dh, dw, LINEAR))) > 8               # the feeder uploads such frames whole


def test_copy_plan_of_irregular_lists():
    rng = np.random.default_rng(3)
    for _ in range(200):
        h = int(rng.integers(1, 200))
        rows = np.flatnonzero(rng.random(h) < rng.random()).astype(np.int32)
        for packed in (True, False):
            plan = copy_plan(rows, packed)
            assert sorted(expand(plan)) == list(rows)
    assert len(copy_plan(np.zeros(0, np.int32))) == 0
    assert copy_plan(np.arange(50, dtype=np.int32)).tolist() == [[0, 50, 0, 1]]
    n = ctypes.c_int(0)
    bad = np.array([4, 4], np.int32)
    assert _native.load().psd_upload_rows_plan(bad.ctypes.data, 2, 1, None, 0, ctypes.addressof(n)) == _native.PSD_ERR_INVALID


class _HostBuffer:
    """DeviceBuffer stand-in in host memory; starts poisoned like fresh device memory."""

    def __init__(self, nbytes):
        self.mem = np.full(nbytes, 0xA5, np.uint8)
        self.nbytes = nbytes
        self.ptr = self.mem.ctypes.data
        self.row_calls = self.full_calls = 0

    def upload_unordered(self, host, offset=0):
        self.full_calls += 1
        self.mem[offset: offset + host.nbytes] = host.reshape(-1)

    def upload_rows(self, frame, offset, rows):
        self.row_calls += 1
        h, w, c = frame.shape
        view = self.mem[offset: offset + h * w * c].reshape(h, w * c)
        view[rows] = frame.reshape(h, w * c)[rows]

    def free(self):
        pass


class _HostDeviceEngine:
    """The part of ScoringEngine the feeder and SceneManager use, over host memory and the oracle: `analyze_device`
    reads the frames back out of the poisoned buffers, so a row the feeder failed to upload shows in the results."""

    def __init__(self, oracle_engine):
        self.oracle = oracle_engine
        self.buffers = []

    def alloc(self, nbytes):
        self.buffers.append(_HostBuffer(nbytes))
        return self.buffers[-1]

    def _view(self, ptr, n, h, w, stride):
        for b in self.buffers:
            if b.ptr <= ptr < b.ptr + b.nbytes:
                off = ptr - b.ptr
                return np.stack([b.mem[off + t * stride: off + t * stride + h * w * 3].reshape(h, w, 3) for t in range(n)])
        raise AssertionError("unknown device pointer")

    def copy_d2d(self, dst, src, nbytes):
        ctypes.memmove(dst, src, nbytes)

    def synchronize(self):
        pass

    def analyze_device(self, d_frames, n, height, width, frame_stride, d_prev=None, flags=0, edge_kernels=(0,), downscale=1.0,
                       hash_sizes=(), interpolation=1, want_frames=False):
        frames = self._view(d_frames, n, height, width, frame_stride)
        prev = self._view(d_prev, 1, height, width, frame_stride)[0] if d_prev else None
        kw = {"downscale": downscale, "interpolation": interpolation} if downscale > 1.0 else {}
        out = {"records": None, "edge_xor": {}, "thumbs": {}, "frames": None}
        out["size"] = (max(1, round(height / downscale)), max(1, round(width / downscale))) if downscale > 1.0 else (height, width)
        if flags:
            out["records"] = self.oracle.score_host(frames, prev=prev, flags=flags, edge_kernel=list(edge_kernels)[0], **kw)
            if flags & 8:
                out["edge_xor"][list(edge_kernels)[0]] = out["records"]["edge_xor"]
        for size in hash_sizes:
            out["thumbs"][size] = self.oracle.hash_thumbs_host(frames, size, **kw)
        if want_frames and downscale > 1.0:
            out["frames"] = self.oracle.downscale_host(frames, downscale, interpolation)
        return out


class _HostDeviceEngineWithRows(TapRowPolicy, _HostDeviceEngine):
    """... with the product's row policy over the product's row list and copy plan."""

    def downscale_source_rows(self, h, w, dh, dw, interp=1):
        return source_rows(h, w, dh, dw, interp)

    def upload_rows_plan(self, rows, packed=True):
        return copy_plan(rows, packed)


def _run(engine, frames, factor, interpolation, detector, callback=None):
    sm = psd.SceneManager(engine=engine, batch_frames=16)
    sm.auto_downscale = False
    sm.downscale = factor
    sm.interpolation = interpolation
    sm.add_detector(detector)
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), callback=callback)
    return [c.frame_num for c in sm.get_cut_list(show_warning=False)]


@pytest.mark.parametrize("interpolation", [psd.Interpolation.LINEAR, psd.Interpolation.NEAREST, psd.Interpolation.AREA])
def test_scene_manager_over_a_feeder_that_uploads_tap_rows_only(golden, oracle_engine, interpolation):
    frames = golden_clip(golden, "scenes_a")
    factor = 4      # 72 x 128 frames: 36 of 72 rows carry bilinear taps
    shown_a, shown_b = [], []
    by_rows = _HostDeviceEngineWithRows(oracle_engine)
    whole = _HostDeviceEngine(oracle_engine)
    cuts_a = _run(by_rows, frames, factor, interpolation, psd.ContentDetector(engine=by_rows),
                  callback=lambda img, pos: shown_a.append((pos.frame_num, int(img.sum()))))
    cuts_b = _run(whole, frames, factor, interpolation, psd.ContentDetector(engine=whole),
                  callback=lambda img, pos: shown_b.append((pos.frame_num, int(img.sum()))))
    assert cuts_a == cuts_b and len(cuts_a) > 0 and shown_a == shown_b
    rows_used = sum(b.row_calls for b in by_rows.buffers)
    if interpolation == psd.Interpolation.AREA:        # every row carries weight: plain uploads
        assert rows_used == 0 and sum(b.full_calls for b in by_rows.buffers) == len(frames)
    else:
        assert rows_used == len(frames) and sum(b.full_calls for b in by_rows.buffers) == 0
    assert sum(b.row_calls for b in whole.buffers) == 0


def test_row_policy_per_shape():
    class E(TapRowPolicy):
        calls = 0

        def downscale_source_rows(self, h, w, dh, dw, interp=1):
            E.calls += 1
            return source_rows(h, w, dh, dw, interp)

        def upload_rows_plan(self, rows, packed=True):
            return copy_plan(rows, packed)

    e = E()
    rows = e.tap_rows(1080, 1920, 7.0, LINEAR)
    assert rows is not None and len(rows) == 308 and e.tap_rows(1080, 1920, 7.0, LINEAR) is rows and E.calls == 1
    assert len(e.tap_rows(1080, 1920, 7.5, LINEAR)) == 288       # the default pipeline: two copies per frame
    assert e.tap_rows(1080, 1920, 1.0, LINEAR) is None            # no downscale: every row is scored
    assert e.tap_rows(270, 481, 1.5, LINEAR) is None              # most rows carry taps
    assert e.tap_rows(1080, 1920, 7.0, AREA) is None              # every row carries weight
    assert len(e.tap_rows(1080, 1920, 7.0, 2)) == 4 * 154         # CUBIC: four of every seven rows
    assert len(e.tap_rows(1080, 1920, 7.5, 2)) == 4 * 144         # ... the default pipeline's 7.5: 53 % of the rows
    assert e.tap_rows(1080, 1920, 7.0, 5) is None                 # no such filter: the scoring call reports it
    assert e.tap_rows(1080, 1920, 4.3, LINEAR) is None            # ~70 copies per frame: not worth it
    # the feeder asks the engine; an engine without the policy gets whole frames
    assert len(_DeviceFeeder(e, 4, factor=7.5, interpolation=LINEAR)._rows_of(1080, 1920)) == 288
    assert _DeviceFeeder(e, 4, factor=1.0)._rows_of(1080, 1920) is None
    assert _DeviceFeeder(object(), 4, factor=7.0)._rows_of(1080, 1920) is None


class _AsyncHostBuffer(_HostBuffer):
    """... whose batched row uploads are ASYNCHRONOUS like psd_upload_rows_batch: the rows reach the buffer only when the
    engine's fence runs, so a feeder that forgets to flush or to fence a batch scores poisoned rows."""

    def __init__(self, nbytes, engine):
        super().__init__(nbytes)
        self._engine = engine
        self.batch_calls, self.batch_sizes = 0, []

    def upload_rows_batch(self, frames, offset, rows, frame_stride):
        self.batch_calls += 1
        self.batch_sizes.append(len(frames))
        for i, f in enumerate(frames):
            f = np.array(f, copy=True)                   # (the host frames are free when the call returns)
            self._engine.in_flight.append((self, offset + i * frame_stride, f, np.array(rows)))


class _BatchingEngine(_HostDeviceEngineWithRows):
    def __init__(self, oracle_engine):
        super().__init__(oracle_engine)
        self.in_flight, self.fences = [], 0

    def alloc(self, nbytes):
        self.buffers.append(_AsyncHostBuffer(nbytes, self))
        return self.buffers[-1]

    def upload_fence(self, wait_on_host=False):
        self.fences += 1
        for buf, off, frame, rows in self.in_flight:
            h, w, c = frame.shape
            buf.mem[off: off + h * w * c].reshape(h, w * c)[rows] = frame.reshape(h, w * c)[rows]
        self.in_flight = []

    def analyze_device(self, d_frames, *a, **k):
        # (the decode thread may already be filling the NEXT buffer: only this batch's own buffer must be settled)
        mine = next(b for b in self.buffers if b.ptr <= d_frames < b.ptr + b.nbytes)
        assert not [x for x in list(self.in_flight) if x[0] is mine], "a batch was scored before its uploads were fenced"
        return super().analyze_device(d_frames, *a, **k)


@pytest.mark.parametrize("batch_frames,feed_batch", [(16, 16), (16, 5), (7, 16), (64, 3)])
def test_feeder_batches_row_uploads_and_fences_every_batch(golden, oracle_engine, batch_frames, feed_batch):
    """The feeder's protocol over `upload_rows_batch` (ABI 5): frames are handed over `FEED_BATCH` at a time, what is pending
    when a batch of the SceneManager closes is flushed, and one fence per batch orders the scoring behind the copies -- same
    cuts and callback frames as whole-frame uploads."""
    frames = golden_clip(golden, "scenes_a")
    keep = _DeviceFeeder.FEED_BATCH
    _DeviceFeeder.FEED_BATCH = feed_batch
    try:
        eng, whole = _BatchingEngine(oracle_engine), _HostDeviceEngine(oracle_engine)

        def run(engine, shown):
            sm = psd.SceneManager(engine=engine, batch_frames=batch_frames)
            sm.auto_downscale = False
            sm.downscale = 4
            sm.add_detector(psd.ContentDetector(engine=engine))
            sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0), callback=lambda img, pos: shown.append((pos.frame_num, int(img.sum()))))
            return [c.frame_num for c in sm.get_cut_list(show_warning=False)]

        shown_a, shown_b = [], []
        assert run(eng, shown_a) == run(whole, shown_b) and shown_a == shown_b and len(shown_a) > 0
    finally:
        _DeviceFeeder.FEED_BATCH = keep
    n_batches = -(-len(frames) // batch_frames)
    assert eng.fences == n_batches and not eng.in_flight
    sizes = [s for b in eng.buffers for s in getattr(b, "batch_sizes", [])]
    assert sum(sizes) == len(frames) and max(sizes) <= min(feed_batch, batch_frames)
    assert sum(b.row_calls + b.full_calls for b in eng.buffers) == 0      # nothing went frame by frame


def test_random_shapes_rows_not_listed_never_reach_the_resized_frame():
    """A seeded slice of a randomized sweep (91 k cases in 60 s, none bad): source sizes 1..299 per axis, destination sizes from a common
    factor, independent per axis, enlarging, or an integer divisor; NEAREST / LINEAR / AREA."""
    import cv2  # the oracle's shim

    for case in range(400):
        rng = np.random.default_rng([77, case])
        h, w = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        k = int(rng.integers(0, 4))
        if k == 0:
            f = rng.uniform(1.0, 9.0)
            dh, dw = max(1, round(h / f)), max(1, round(w / f))
        elif k == 1:
            dh, dw = int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1))
        elif k == 2:
            dh, dw = int(rng.integers(1, 2 * h + 2)), int(rng.integers(1, 2 * w + 2))
        else:
            f = int(rng.integers(1, 6))
            dh, dw = max(1, h // f), max(1, w // f)
        interp = int(rng.choice([NEAREST, LINEAR, AREA]))
        rows = source_rows(h, w, dh, dw, interp)
        assert len(rows) and np.all(np.diff(rows) > 0) and rows[0] >= 0 and rows[-1] < h, (h, w, dh, dw, interp)
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = cv2.resize(frame, (dw, dh), interpolation=interp)
        for poison in (rng.integers(0, 256, (h, w, 3), dtype=np.uint8), np.full((h, w, 3), 255 if frame.mean() < 128 else 0, np.uint8)):
            poison[rows] = frame[rows]
            assert np.array_equal(cv2.resize(poison, (dw, dh), interpolation=interp), want), (h, w, dh, dw, interp)
