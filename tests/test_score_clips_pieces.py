"""`ScoringEngine.score_clips` cuts long resident runs into pipelined pieces (one frame of overlap where a cut falls inside
a clip).  The bookkeeping is host logic: exercised here without a GPU, with the engine's two native calls replaced by the
CPU oracle reading the same "device" addresses."""
import ctypes

import numpy as np
import pytest

from pyscenedetect_amd import _native
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.engine import _plan_pieces


class FakeDeviceClip:
    """A slice of a numpy pool that looks like a device tensor (shape + data_ptr)."""

    def __init__(self, arr):
        self.arr, self.shape = arr, arr.shape

    def data_ptr(self):
        return self.arr.ctypes.data


class OracleBackedEngine(E.ScoringEngine):
    def __init__(self):      # no native engine
        import threading

        self._lock = threading.RLock()
        self.kernel_ms_acc = 0.0
        self.pending, self.submitted, self.max_pending = [], [], 0

    def close(self):
        pass

    def submit_device_segments(self, d_frames, n, height, width, seg_first, row_stride=None, frame_stride=None, flags=7, edge_kernel=0, stream=None):
        from oracle import lib as orc

        assert len(self.pending) < _native.MAX_INFLIGHT
        frames = np.ctypeslib.as_array(ctypes.cast(d_frames, ctypes.POINTER(ctypes.c_uint8)), (n, height, width, 3))
        seg = list(seg_first) + [n]
        assert seg[0] == 0
        recs = np.concatenate([orc.score_batch(frames[a:b], None, flags=flags & 7) for a, b in zip(seg[:-1], seg[1:])])
        self.pending.append(recs)
        self.submitted.append((n, list(seg_first)))
        self.max_pending = max(self.max_pending, len(self.pending))

    def collect(self, n, sums_only=False):
        recs = self.pending.pop(0)
        assert len(recs) == n
        return E._sums_of(recs) if sums_only else recs


def test_plan_pieces(monkeypatch):
    fb = 1 << 20                                      # 1 MiB frames
    monkeypatch.delenv("PSD_CLIPS_TAIL_MB", raising=False)
    assert _plan_pieces(3000, [0], fb) == [0, 3000]                     # 3 GB: shorter than three tails of 1.5 GiB
    assert _plan_pieces(20000, [0], fb) == [0, 16000, 20000]            # a fifth of the run
    assert _plan_pieces(20000, [0], fb, last_run=False) == [0, 20000]   # only the last run of a call is cut
    assert _plan_pieces(20000, [0, 15000, 17000], fb) == [0, 15000, 20000]      # the cut snaps to the nearest clip start nearby ...
    assert _plan_pieces(20000, [0, 9000], fb) == [0, 16000, 20000]      # ... not to a far one
    assert _plan_pieces(100000, [0], fb) == [0, 100000 - 6144, 100000]  # at most 6 GiB
    monkeypatch.setenv("PSD_CLIPS_TAIL_MB", "0")
    assert _plan_pieces(20000, [0], fb) == [0, 20000]
    monkeypatch.setenv("PSD_CLIPS_TAIL_MB", "16")
    assert _plan_pieces(100, [0, 29, 70], fb) == [0, 84, 100]


@pytest.mark.parametrize("sums_only", [False, True])
@pytest.mark.parametrize("tail_mb", ["1", "0"])
def test_pieces_equal_per_clip_scores(monkeypatch, sums_only, tail_mb):
    from oracle import lib as orc

    monkeypatch.setenv("PSD_CLIPS_TAIL_MB", tail_mb)
    rng = np.random.default_rng(5)
    h, w = 96, 128                                   # 36 KiB per frame: 28 frames per 1 MiB piece
    lens = [5, 61, 1, 30, 117, 2, 40]
    pool = rng.integers(0, 256, (sum(lens), h, w, 3), dtype=np.uint8)
    clips, off = [], 0
    for n in lens:
        clips.append(FakeDeviceClip(pool[off:off + n]))
        off += n
    other = FakeDeviceClip(rng.integers(0, 256, (700, 40, 56, 3), dtype=np.uint8))    # a second resolution, its own (the last) run
    order = []
    eng = OracleBackedEngine()
    got = eng.score_clips(clips + [other], flags=7, sums_only=sums_only, on_ready=lambda i, r: order.append(i))
    for c, g in zip(clips + [other], got):
        want = orc.score_batch(c.arr, None, flags=7)
        for f in g.dtype.names:
            assert np.array_equal(g[f], want[f]), f
    assert sorted(order) == list(range(len(lens) + 1))
    assert order[: len(lens)] == list(range(len(lens)))          # clips of a run become ready in order
    if tail_mb == "1":
        # the last run (the second resolution: 700 frames of 6.6 KiB, a tail of 156) is the one with a tail piece
        assert len(eng.submitted) == 3 and eng.max_pending == 3
    else:
        assert len(eng.submitted) == 2


def test_tail_piece_inside_a_clip(monkeypatch):
    """The only run is one long clip: the tail piece starts inside it, one frame early, and that frame's record is dropped."""
    from oracle import lib as orc

    monkeypatch.setenv("PSD_CLIPS_TAIL_MB", "1")
    rng = np.random.default_rng(8)
    pool = rng.integers(0, 256, (150, 96, 128, 3), dtype=np.uint8)      # 36 KiB per frame: a tail of 28 frames
    eng = OracleBackedEngine()
    got = eng.score_clips([FakeDeviceClip(pool)], flags=7, on_ready=lambda i, r: None)
    assert len(OracleBackedEngine().score_clips([FakeDeviceClip(pool)], flags=7)) == 1     # (nobody to decide meanwhile: no cut)
    want = orc.score_batch(pool, None, flags=7)
    for f in want.dtype.names:
        assert np.array_equal(got[0][f], want[f]), f
    assert eng.submitted == [(122, [0]), (29, [0])]


def test_error_in_on_ready_retires_pending(monkeypatch):
    monkeypatch.setenv("PSD_CLIPS_TAIL_MB", "1")
    rng = np.random.default_rng(6)
    pool = rng.integers(0, 256, (200, 96, 128, 3), dtype=np.uint8)
    eng = OracleBackedEngine()

    def boom(i, r):
        raise RuntimeError("decision failed")

    with pytest.raises(RuntimeError, match="decision failed"):
        eng.score_clips([FakeDeviceClip(pool[:120]), FakeDeviceClip(pool[120:])], on_ready=boom)
    assert eng.pending == []


def test_default_rule_cuts_only_with_enough_clips_ahead(monkeypatch):
    """Without the override a tail piece needs about six clips in front of it and 512 frames of its own."""
    monkeypatch.delenv("PSD_CLIPS_TAIL_MB", raising=False)
    monkeypatch.setattr(E, "_plan_pieces", lambda total, first, fb, last_run=True: [0, total - 600, total] if last_run and total > 1800 else [0, total])
    rng = np.random.default_rng(9)
    pool = rng.integers(0, 256, (2400, 4, 8, 3), dtype=np.uint8)
    few = [FakeDeviceClip(pool[:1200]), FakeDeviceClip(pool[1200:])]
    many = [FakeDeviceClip(pool[i * 240:(i + 1) * 240]) for i in range(10)]
    for clips, pieces in ((few, 1), (many, 2)):
        eng = OracleBackedEngine()
        eng.score_clips(clips, flags=1, on_ready=lambda i, r: None)
        assert len(eng.submitted) == pieces
