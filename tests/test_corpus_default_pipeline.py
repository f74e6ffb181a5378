"""CPU: the packed / sharded flow (``corpus.detect_corpus``: what BASELINE configs[3] and [4] time) computes what the reference computes
per video -- ``detect(video, detector_cls())``, i.e. ``SceneManager`` with ``auto_downscale=True`` resizing every frame to about 256
pixels width in front of the detector (``scene_manager.py:110,123-140,666-678``; ``benchmark/__main__.py:44-61``).

Round-5 review, gap N1: the flow scored full-resolution frames, so its cut lists were not the reference benchmark's.  Here, over the
CPU oracle behind the engine interface: the cut lists and every frame's ``content_val`` of the UNMODIFIED reference's default
pipeline (``tests/golden/corpus_default_pipeline.json``, ``oracle/gen_corpus_golden.py``), the bookkeeping of pieces that start inside
a clip behind a downscale, and -- in the build container -- a slice of the live differential fuzz against the reference itself.
The GPU twin (every clip incl. 1080p and 4K through ``psd_score_segments_downscaled_device``) is ``tests/test_gpu_corpus_default.py``."""
import ctypes
import os
import sys

import numpy as np
import pytest

from pyscenedetect_amd import _native, corpus, epilogue
from pyscenedetect_amd import engine as E
from tests._helpers import CORPUS_DETECTORS, corpus_clip, corpus_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_CLIPS = [k for k, v in corpus_golden()["clips"].items() if v["cpu"]]


def test_downscale_size_is_the_references_arithmetic():
    """``compute_downscale_factor(max(frame_size))`` and ``max(1, round(size / factor))`` (scene_manager.py:123-140,525-528,670-678)."""
    assert E.downscale_size(1080, 1920, "auto") == (7.5, 144, 256)
    assert E.downscale_size(360, 640, "auto") == (2.5, 144, 256)
    assert E.downscale_size(2160, 3840, "auto") == (15.0, 144, 256)
    assert E.downscale_size(640, 360, "auto") == (2.5, 256, 144)          # the factor comes from the larger side
    assert E.downscale_size(120, 200, "auto") == (1.0, 120, 200)          # narrower than 256: untouched
    assert E.downscale_size(144, 256, "auto") == (1.0, 144, 256)          # exactly 256: factor 1.0, not "> 1.0"
    assert E.downscale_size(300, 257, "auto") == (300 / 256.0, 256, 219)
    assert E.downscale_size(10, 4000, "auto") == (15.625, 1, 256)         # max(1, round(0.64))
    assert E.downscale_size(101, 101, 2) == (2.0, 50, 50)                 # Python's round: half to even
    assert E.downscale_size(100, 100, None) == E.downscale_size(100, 100, 1) == E.downscale_size(100, 100, 0.5) == (1.0, 100, 100)
    with pytest.raises(ValueError):
        E.downscale_size(100, 100, "half")
    for name, c in corpus_golden()["clips"].items():
        f, dh, dw = E.downscale_size(c["h"], c["w"], "auto")
        assert [dh, dw] == c["scored_size"] and f == float(c["factor"]), name


def test_default_pipeline_equals_the_reference_goldens(oracle_engine):
    """Cut lists of all four detectors and content_val of every frame, bit for bit, for clips of seven resolutions in ONE call."""
    g = corpus_golden()
    clips = [corpus_clip(k) for k in CPU_CLIPS]
    res = corpus.detect_corpus(oracle_engine, clips, g["fps"], CORPUS_DETECTORS)
    for name, r in zip(CPU_CLIPS, res):
        assert r == g["clips"][name]["cuts"], name
    assert sum(len(r["content"]) for r in res) >= 25
    recs = corpus.score_clips(oracle_engine, clips, _native.SCORE_HSV_SAD, downscale="auto")
    for name, r in zip(CPU_CLIPS, recs):
        c = g["clips"][name]
        assert r.dtype == _native.SUMS_DTYPE
        got = epilogue.content_scores(r, *c["scored_size"])["content_val"]
        want = c["content_val"]
        assert want[0] is None and got[0] == 0.0
        assert got[1:].tolist() == want[1:], name


def test_full_resolution_is_another_computation(oracle_engine):
    """What round 5's flow did (no resize) gives other scores on every clip wider than 256 pixels and other DECISIONS on the noisy
    one -- i.e. the goldens above tell the two apart."""
    g = corpus_golden()
    differ = 0
    for name in ("bbc_c", "noisy_a", "small_a"):
        c = g["clips"][name]
        r = corpus.score_clips(oracle_engine, [corpus_clip(name)], _native.SCORE_HSV_SAD, downscale=None)[0]
        full = epilogue.content_scores(r, c["h"], c["w"])["content_val"][1:].tolist()
        if name == "small_a":
            assert full == c["content_val"][1:]                   # never resized: the same frames either way
        else:
            assert full != c["content_val"][1:], name
            differ += 1
    assert differ == 2
    full = corpus.detect_corpus(oracle_engine, [corpus_clip("noisy_a")], g["fps"], CORPUS_DETECTORS, auto_downscale=False)[0]
    assert full != g["clips"]["noisy_a"]["cuts"]


def test_manual_downscale_and_interpolation(oracle_engine):
    """``auto_downscale=False`` + ``downscale`` / ``interpolation`` are SceneManager's attributes of the same names."""
    import cv2  # the oracle's shim

    clip = corpus_clip("bbc_c")
    for factor, interp in ((3, 1), (2, 0), (4, 3)):
        small = np.stack([cv2.resize(f, (round(640 / factor), round(360 / factor)), interpolation=interp) for f in clip])
        want = corpus.detect_corpus(oracle_engine, [small], 25.0, CORPUS_DETECTORS, auto_downscale=False)
        got = corpus.detect_corpus(oracle_engine, [clip], 25.0, CORPUS_DETECTORS, auto_downscale=False, downscale=factor, interpolation=interp)
        assert got == want, (factor, interp)


class FakeDeviceClip:
    """A slice of a numpy pool that looks like a device tensor (shape + data_ptr)."""

    def __init__(self, arr):
        self.arr, self.shape = arr, arr.shape

    def data_ptr(self):
        return self.arr.ctypes.data


class OracleBackedEngine(E.ScoringEngine):
    """``ScoringEngine.score_clips``'s own bookkeeping over the CPU oracle: the two native submissions it issues are answered from
    the same "device" addresses (tests/test_score_clips_pieces.py's stand-in, with the downscaled entry point)."""

    def __init__(self):      # no native engine
        import threading

        self._lock = threading.RLock()
        self.kernel_ms_acc = 0.0
        self.pending, self.calls = [], []

    def close(self):
        pass

    def _score(self, frames, seg_first, flags, size=None, interpolation=1):
        from oracle import lib as orc

        if size is not None:
            import cv2  # the oracle's shim

            frames = np.stack([cv2.resize(f, (size[1], size[0]), interpolation=interpolation) for f in frames])
        seg = list(seg_first) + [len(frames)]
        assert seg[0] == 0
        self.pending.append(np.concatenate([orc.score_batch(frames[a:b], None, flags=flags & 7) for a, b in zip(seg[:-1], seg[1:])]))

    def submit_device_segments(self, d_frames, n, height, width, seg_first, row_stride=None, frame_stride=None, flags=7, edge_kernel=0, stream=None):
        assert len(self.pending) < _native.MAX_INFLIGHT
        frames = np.ctypeslib.as_array(ctypes.cast(d_frames, ctypes.POINTER(ctypes.c_uint8)), (n, height, width, 3))
        self.calls.append(("full", n, height, width))
        self._score(frames, seg_first, flags)

    def submit_device_segments_downscaled(self, d_frames, n, src_h, src_w, dst_h, dst_w, seg_first, frame_stride=None, flags=1, edge_kernel=0,
                                          interpolation=1, stream=None):
        assert len(self.pending) < _native.MAX_INFLIGHT
        frames = np.ctypeslib.as_array(ctypes.cast(d_frames, ctypes.POINTER(ctypes.c_uint8)), (n, src_h, src_w, 3))
        self.calls.append(("small", n, src_h, src_w, dst_h, dst_w, interpolation))
        self._score(frames, seg_first, flags, (dst_h, dst_w), interpolation)

    def collect(self, n, sums_only=False):
        recs = self.pending.pop(0)
        assert len(recs) == n
        return E._sums_of(recs) if sums_only else recs


@pytest.mark.parametrize("tail_mb", ["1", "0"])
def test_pieces_behind_the_downscale_equal_per_clip_scores(monkeypatch, oracle_engine, tail_mb):
    """Resident clips of two resolutions (one resized by 1.25 / 1.5, one below 256 pixels), the last run cut into pieces that start inside a
    clip: every clip's records == the oracle engine's for that clip alone behind the same downscale; each resolution took its own
    entry point with its own target size."""
    monkeypatch.setenv("PSD_CLIPS_TAIL_MB", tail_mb)
    rng = np.random.default_rng(8)
    wide = rng.integers(0, 256, (5 + 61 + 1 + 30 + 117, 96, 320, 3), dtype=np.uint8)      # 90 KiB per frame: 11 frames per 1 MiB tail
    narrow = rng.integers(0, 256, (23 + 40, 48, 200, 3), dtype=np.uint8)
    clips, off = [], 0
    for n in (5, 61, 1, 30, 117):
        clips.append(FakeDeviceClip(wide[off:off + n]))
        off += n
    small = [FakeDeviceClip(narrow[:23]), FakeDeviceClip(narrow[23:])]
    order = [small[0], clips[0], clips[1], small[1]] + clips[2:]
    for ds, interp in (("auto", 1), (1.5, 0)):
        eng = OracleBackedEngine()
        done = []
        got = eng.score_clips(order, flags=7, on_ready=lambda i, r: done.append(i), downscale=ds, interpolation=interp)
        assert sorted(done) == list(range(len(order)))
        f_wide, dh, dw = E.downscale_size(96, 320, ds)
        f_narrow = E.downscale_size(48, 200, ds)[0]
        assert (f_wide, f_narrow) == ((1.25, 1.0) if ds == "auto" else (1.5, 1.5))
        for c, r in zip(order, got):
            f = f_wide if c.shape[2] == 320 else f_narrow
            want = oracle_engine.score_host(c.arr, flags=7, downscale=f, interpolation=interp)
            assert r.tobytes() == want.tobytes()
        kinds = {call[0] for call in eng.calls}
        assert kinds == ({"small", "full"} if ds == "auto" else {"small"})
        assert all(call[4:] == (dh, dw, interp) for call in eng.calls if call[0] == "small" and call[3] == 320)
        if tail_mb == "1":
            assert len([call for call in eng.calls if call[3] == 320]) == 2          # the last run was cut once


def test_host_clips_and_engines_without_a_downscale(oracle_engine):
    """Clips in host memory and the engines that cannot pack (clip by clip) give the same records; an engine whose ``score_clips`` knows
    no downscale still packs a corpus in which nothing is resized, and is driven clip by clip where something is."""
    clips = [corpus_clip("bbc_c"), corpus_clip("small_a"), corpus_clip("edge_a")]

    class Packing:
        def __init__(self):
            self.packed = 0

        def score_clips(self, cl, flags, edge_kernel=0, sums_only=False):
            self.packed += 1
            return [E._sums_of(oracle_engine.score_host(c, flags=flags)) if sums_only else oracle_engine.score_host(c, flags=flags) for c in cl]

        score_host = staticmethod(oracle_engine.score_host)

    want = corpus.score_clips(oracle_engine, clips, 7, downscale="auto")
    p = Packing()
    got = corpus.score_clips(p, clips, 7, downscale="auto")
    assert p.packed == 0 and all(a.tobytes() == b.tobytes() for a, b in zip(got, want))
    got = corpus.score_clips(p, clips[1:2], 7, downscale="auto")                         # nothing to resize: packed as before
    assert p.packed == 1 and got[0].tobytes() == want[1].tobytes()


@pytest.mark.skipif(not os.path.isdir("/root/reference/scenedetect"), reason="the reference checkout is only in the build container")
def test_live_reference_behind_its_resize_slice():
    """``tools/fuzz_epilogue_vs_reference.py --downscale``: the unmodified reference's SceneManager with its default auto-downscale (or a
    manual factor; LINEAR / NEAREST / AREA) against ``detect_corpus`` over the oracle engine with the same setting, random clips,
    detectors and parameters incl. the edge term (its dilation kernel sized by the RESIZED frame)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import logging
    import warnings

    import fuzz_epilogue_vs_reference as X

    level = logging.root.manager.disable
    logging.disable(logging.CRITICAL)
    resized = 0
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for case_no in range(150):
                rng = np.random.default_rng([61, case_no])
                frames, fps, name, kw, kernel = X.draw(rng)
                while name == "hash":
                    frames, fps, name, kw, kernel = X.draw(rng)
                resize = X.draw_downscale(rng)
                a = X.F.decisions(X.F.outcome(lambda: {"cuts": X.reference_cuts(frames, fps, name, kw, kernel, resize)}))
                b = X.F.decisions(X.F.outcome(lambda: {"cuts": X.corpus_cuts(frames, fps, name, kw, kernel, resize)}))
                assert a == b, (case_no, list(frames.shape), name, kw, kernel, resize, a, b)
                ds = "auto" if resize["auto_downscale"] else resize.get("downscale", 1)
                resized += E.downscale_size(frames.shape[1], frames.shape[2], ds)[0] > 1.0
    finally:
        logging.disable(level)
    assert resized >= 30


def _bbc_layout(tmp_path, names):
    """The golden clips `names` as a dataset in the BBC layout (benchmark/dataset.py:77-106), annotations = the generator's shot starts."""
    os.makedirs(tmp_path / "videos")
    os.makedirs(tmp_path / "fixed")
    for i, name in enumerate(names):
        frames = corpus_clip(name)
        np.save(tmp_path / "videos" / f"bbc_{i + 1:02d}.npy", frames)
        bounds = [0, *corpus_golden()["clips"][name]["shot_starts"], len(frames)]
        with open(tmp_path / "fixed" / f"{i + 1:02d}-scenes.txt", "w") as f:
            for a, b in zip(bounds[:-1], bounds[1:]):
                f.write(f"{a}\t{b - 1}\n")


def test_benchmark_harness_packed_equals_one_manager_per_video(tmp_path, oracle_engine):
    """SURVEY 8 f2 meets the packed flow: ``tools/bbc_harness.run_predictions`` is the reference's ``_run_predictions``
    (benchmark/__main__.py:44-61: one default detector per video through a default SceneManager); ``run_predictions_packed`` hands the
    whole dataset to ``detect_corpus`` -- and returns the same predictions, which are also the reference's own (the goldens)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bbc_harness as H

    names = ["bbc_a", "bbc_b", "bbc_c", "noisy_a"]
    _bbc_layout(tmp_path, names)
    samples = H.bbc_samples(str(tmp_path))
    for det, key in H.PACKED.items():
        one = H.run_predictions(samples, det, engine=oracle_engine)
        packed = H.run_predictions_packed(samples, det, oracle_engine)
        assert [r["predicted_cuts"] for r in packed] == [r["predicted_cuts"] for r in one], det
        # (the goldens' adaptive line spells out window_width = 2, min_content_val = 15: the constructor's defaults)
        for name, r in zip(names, packed):
            c = corpus_golden()["clips"][name]
            cuts = c["cuts"][key]
            assert r["predicted_cuts"] == (cuts + [c["n"]] if cuts else []), (det, name)


def test_empty_and_one_frame_clips(oracle_engine):
    """A corpus may hold clips without frames and clips of one frame, behind the downscale as without it."""
    rng = np.random.default_rng(0)
    clips = [np.zeros((0, 360, 640, 3), np.uint8), rng.integers(0, 256, (3, 360, 640, 3), dtype=np.uint8),
             rng.integers(0, 256, (1, 100, 100, 3), dtype=np.uint8)]
    assert corpus.detect_corpus(oracle_engine, [], 25.0, CORPUS_DETECTORS) == []
    for auto in (True, False):
        res = corpus.detect_corpus(oracle_engine, clips, [25.0, 30.0, 24.0], CORPUS_DETECTORS, auto_downscale=auto)
        assert res[0] == res[2] == {"content": [], "adaptive": [], "hist": [], "threshold": []} and set(res[1]) == set(CORPUS_DETECTORS)


def test_decisions_on_worker_threads_equal_the_inline_ones(monkeypatch):
    """``detect_corpus`` decides clips on worker threads when a HistogramDetector is among the detectors (its native epilogue is most of
    the host time of a pass behind the default downscale) and ``psd_epilogue_hist_cuts`` splits long clips over native threads: the same
    cut lists and ``hist_diff`` values, to the bit, as one thread does inline."""
    import numpy as np

    from pyscenedetect_amd import _native, corpus, epilogue

    rng = np.random.default_rng(11)
    n = 5000
    recs = np.zeros(n, _native.RECORD_DTYPE)
    base = rng.integers(0, 400, (8, 256))
    for t in range(n):
        recs["hist"][t] = base[(t // 700) % 8] + rng.integers(0, 6, 256)          # shots of 700 frames with a little noise
    recs["byte_sum"] = recs["hist"].sum(axis=1) * 3
    runs = {}
    for threads in ("1", "2", "5"):
        # (the library reads PSD_EPILOGUE_THREADS once per process: each setting in its own interpreter)
        import json
        import os
        import subprocess
        import sys

        code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from pyscenedetect_amd import epilogue, _native;"
                "recs = np.load(sys.argv[1]); c, d = epilogue.hist_cuts(recs, 25.0); print(json.dumps([c, np.nan_to_num(d, nan=-1.0).tobytes().hex()]))"
                % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "psd_hist_recs_%d.npy" % os.getpid())
        np.save(path, recs)
        out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, env=dict(os.environ, PSD_EPILOGUE_THREADS=threads))
        os.unlink(path)
        assert out.returncode == 0, out.stderr[-800:]
        runs[threads] = json.loads(out.stdout.strip().splitlines()[-1])
    assert runs["1"] == runs["2"] == runs["5"] and len(runs["1"][0]) >= 6
    # the Python-level pool: a stand-in engine that hands out prepared records
    class Eng:
        def score_host(self, frames, prev=None, flags=7, edge_kernel=0, downscale=1.0, interpolation=1):
            return recs[: len(frames)]
    clips = [np.zeros((k, 8, 8, 3), np.uint8) for k in (4000, 1200, 0, 1, 2600)]
    spec = {"hist": {}, "threshold": {}}
    monkeypatch.setenv("PSD_DECIDE_THREADS", "0")
    inline = corpus.detect_corpus(Eng(), clips, 25.0, spec, auto_downscale=False)
    monkeypatch.setenv("PSD_DECIDE_THREADS", "4")
    pooled = corpus.detect_corpus(Eng(), clips, 25.0, spec, auto_downscale=False)
    assert pooled == inline and inline[0]["hist"] and corpus._POOL is not None
