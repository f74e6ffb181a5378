"""CPU: psd_epilogue_hist_cuts_from_diff is the decision loop of psd_epilogue_hist_cuts (histogram_detector.py:98-116) over hist_diff values
computed elsewhere (on the GPU: psd_hist_diff_device, tests/test_gpu_hist_diff.py), and corpus.decide takes records that carry them."""
import numpy as np
import pytest

from pyscenedetect_amd import _native, corpus, epilogue
from pyscenedetect_amd import engine as E


def _records(n, seed):
    rng = np.random.default_rng(seed)
    npix = 144 * 256
    recs = np.zeros(n, E.RECORD_DTYPE)
    base = rng.multinomial(npix, rng.dirichlet(np.ones(256) * 0.3))
    h = np.empty((n, 256), np.uint32)
    for t in range(n):
        if t % 23 == 0:
            base = rng.multinomial(npix, rng.dirichlet(np.ones(256) * 0.3))
        h[t] = base if t % 5 else rng.multinomial(npix, (base + 1.0) / (base + 1.0).sum())
    h[40:50] = 0
    h[40:50, 7] = npix                      # constant frames: the correlation's denominator vanishes -> 1.0
    recs["hist"] = h
    for name in ("sad_h", "sad_s", "sad_v"):
        recs[name] = rng.integers(0, npix * 40, n)
    recs["byte_sum"] = rng.integers(0, npix * 3 * 255, n)
    return recs


@pytest.mark.parametrize("threshold", [0.2, 0.05, 0.6, 1.5, -1.0])
@pytest.mark.parametrize("min_scene_len", [15, 1, 0, 0.5, "00:00:01.000", "12"])
def test_cuts_from_diff_equal_cuts_from_histograms(threshold, min_scene_len):
    recs = _records(300, 3)
    for bins in (128, 256, 100):
        cuts, diff = epilogue.hist_cuts(recs, 25.0, threshold, bins, min_scene_len)
        assert np.isnan(diff[0]) and not np.isnan(diff[1:]).any()
        assert epilogue.hist_cuts_from_diff(diff, 25.0, threshold, min_scene_len) == cuts
        assert epilogue.hist_cuts_from_diff(diff, 30000 / 1001, threshold, min_scene_len, first_frame=100) == \
            epilogue.hist_cuts(recs, 30000 / 1001, threshold, bins, min_scene_len, first_frame=100)[0]
    assert epilogue.hist_cuts_from_diff(np.zeros(0), 25.0) == []


def test_decide_takes_records_that_carry_hist_diff():
    recs = _records(200, 9)
    dets = {"content": {}, "adaptive": {}, "hist": {"threshold": 0.3, "bins": 64}, "threshold": {}}
    want = corpus.decide(recs, 144, 256, 25.0, dets)
    small = np.empty(len(recs), _native.SUMS_DIFF_DTYPE)
    for name in _native.SUMS_DTYPE.names:
        small[name] = recs[name]
    small["hist_diff"] = epilogue.hist_cuts(recs, 25.0, 0.3, 64, 15)[1]
    assert corpus.decide(small, 144, 256, 25.0, dets) == want and len(want["hist"]) > 0


def test_invalid_arguments():
    lib = _native.load()
    with pytest.raises(ValueError):
        _native.check(lib.psd_epilogue_hist_cuts_from_diff(None, 3, 0, 25, 1, None, None, None))
    with pytest.raises(ValueError):
        _native.check(lib.psd_hist_diff_device(None, None, 3, 128, None, None))
