"""GPU: HistogramDetector's hist_diff computed on the device from records that are still in HBM (psd_hist_diff_device) is, bit for bit, what
the host epilogue computes from the histograms (psd_epilogue_hist_cuts: histogram_detector.py:98,156-163 restated) -- and the packed flow that
uses it (score_clips(hist_diff_bins=), detect_corpus) decides what the flow over full records decides."""
import numpy as np
import pytest

from pyscenedetect_amd import _native, corpus, epilogue
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.synth import make_clip

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _device_records(engine, frames, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM):
    import torch

    x = torch.from_numpy(frames).cuda()
    n, h, w, _ = frames.shape
    engine.submit_device(x.data_ptr(), n, h, w, flags=flags)
    recs = engine.collect(n)
    ptr, cnt = engine.last_records_device()
    assert cnt == n
    return recs, ptr


@pytest.mark.parametrize("bins", [128, 256, 100, 64, 3, 1, 255])
def test_device_hist_diff_equals_the_host_epilogue(hip_engine, bins):
    frames, _ = make_clip(31, 90, 72, 128, shot_len=(7, 20))
    frames[20:26] = 0                                   # black frames: one bin, the correlation's denominator vanishes
    frames[26:29] = 255
    frames[40] = np.random.default_rng(1).integers(0, 256, frames[40].shape, dtype=np.uint8)
    recs, ptr = _device_records(hip_engine, frames)
    want = epilogue.hist_cuts(recs, 25.0, 0.2, bins, 15)[1]
    got = hip_engine.hist_diff_device(ptr, len(frames), bins)
    assert np.isnan(got[0]) and np.array_equal(_bits(got[1:]), _bits(want[1:])), np.flatnonzero(_bits(got[1:]) != _bits(want[1:]))[:8]
    assert (want[21:26] == 1.0).all() and (bins < 64 or np.unique(want[1:]).size > 30)


def test_device_hist_diff_large_frames_and_many_records(hip_engine):
    """1080p histograms (bins of up to 2 M pixels: float32 holds them exactly; their squares need the float64 sums) and a long batch."""
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (12, 1080, 1920, 3), dtype=np.uint8)
    frames[3:6] //= 4
    frames[6] = 77
    recs, ptr = _device_records(hip_engine, frames)
    for bins in (128, 256):
        want = epilogue.hist_cuts(recs, 25.0, 0.2, bins, 15)[1]
        got = hip_engine.hist_diff_device(ptr, len(frames), bins)
        assert np.array_equal(_bits(got[1:]), _bits(want[1:]))
    small = rng.integers(0, 256, (5000, 16, 32, 3), dtype=np.uint8)
    small[2500:] //= 3
    recs, ptr = _device_records(hip_engine, small)
    got = hip_engine.hist_diff_device(ptr, len(small), 128)
    assert np.array_equal(_bits(got[1:]), _bits(epilogue.hist_cuts(recs, 25.0, 0.2, 128, 15)[1][1:]))
    assert hip_engine.hist_diff_device(ptr, 0, 128).size == 0
    with pytest.raises(ValueError):
        hip_engine.hist_diff_device(ptr, 5, 0)
    with pytest.raises(ValueError):
        hip_engine.hist_diff_device(ptr, 5, 257)


def test_packed_clips_with_hist_diff_from_the_device(hip_engine):
    """score_clips(hist_diff_bins=): resident clips of two resolutions packed per resolution, behind the default downscale and at full
    resolution -- sums identical to the full records', hist_diff identical to the host epilogue's per clip (NaN at every clip's first frame),
    and detect_corpus (which asks for it) decides what the decisions over full records are."""
    import torch

    host = [make_clip(11, 70, 180, 320, shot_len=(9, 25))[0], make_clip(12, 45, 180, 320, shot_len=(9, 25))[0],
            make_clip(13, 33, 360, 640, shot_len=(9, 25))[0], make_clip(14, 1, 180, 320)[0]]
    pool_a = torch.from_numpy(np.concatenate([host[0], host[1], host[3]])).cuda()
    pool_b = torch.from_numpy(host[2]).cuda()
    dev = [pool_a[:70], pool_a[70:115], pool_b, pool_a[115:116]]
    flags = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
    for ds in ("auto", None, 1.5):
        full = hip_engine.score_clips(dev, flags=flags, downscale=ds)
        for bins in (128, 200):
            got = hip_engine.score_clips(dev, flags=flags, downscale=ds, hist_diff_bins=bins)
            for g, f in zip(got, full):
                assert g.dtype == _native.SUMS_DIFF_DTYPE and len(g) == len(f)
                for name in _native.SUMS_DTYPE.names:
                    assert np.array_equal(g[name], f[name]), (ds, name)
                want = epilogue.hist_cuts(f, 25.0, 0.2, bins, 15)[1]
                assert np.isnan(g["hist_diff"][0]) and np.array_equal(_bits(g["hist_diff"][1:]), _bits(want[1:])), (ds, bins)
    spec = {"content": {}, "adaptive": {}, "hist": {"threshold": 0.1}, "threshold": {}}
    got = corpus.detect_corpus(hip_engine, dev, 25.0, spec)
    want = [corpus.decide(hip_engine.score_clips([c], flags=flags, downscale="auto")[0], *corpus.scored_size(c.shape[1], c.shape[2], "auto"), 25.0, spec)
            for c in dev]
    assert got == want and sum(len(r["hist"]) for r in got) >= 3
    # clips in host memory come back as full records and decide the same
    assert corpus.detect_corpus(hip_engine, host, 25.0, spec) == want
