"""CPU: the native whole-clip epilogues (psd_epilogue_*) against the per-frame Python detectors on
random score streams -- frame rates incl. NTSC, min_scene_len as frames / seconds / strings, both
filter modes, all threshold methods.  Both restate the reference; they must agree exactly."""
import numpy as np
import pytest

import pyscenedetect_amd as psd
from pyscenedetect_amd import FlashFilter, FrameTimecode, epilogue
from pyscenedetect_amd._native import RECORD_DTYPE

FPS = [25.0, 29.97, 23.976, 60.0, 12.5]
LENS = [0, 1, 7, 15, 0.3, 0.61, "0.5s", "00:00:00.400", "12"]


def _scores(rng, n):
    base = rng.gamma(1.5, 6.0, n)
    spikes = rng.random(n) < 0.08
    base[spikes] += rng.uniform(20, 200, spikes.sum())
    if rng.random() < 0.3:
        base[rng.integers(0, n, n // 10)] = 0.0
    return base


@pytest.mark.parametrize("seed", range(40))
def test_content_and_adaptive(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 400))
    fps = FPS[seed % len(FPS)]
    length = LENS[int(rng.integers(0, len(LENS)))]
    scores = _scores(rng, n)
    thr = float(rng.uniform(5, 60))
    mode = FlashFilter.Mode.MERGE if seed % 2 else FlashFilter.Mode.SUPPRESS

    class C(psd.ContentDetector):
        def _score_from_record(self, timecode, record, height, width):
            return float(scores[timecode.frame_num])

    det = C(threshold=thr, min_scene_len=length, filter_mode=mode, engine=object())
    want = []
    for i in range(n):
        want += [c.frame_num for c in det.process_record(FrameTimecode(i, fps), None, 1, 1)]
    got = epilogue.content_cuts(scores, fps, thr, length, 0 if mode == FlashFilter.Mode.MERGE else 1)
    assert got == want, (seed, fps, length, mode)

    w = int(rng.integers(1, 5))
    athr, mcv = float(rng.uniform(1.5, 5)), float(rng.uniform(5, 30))

    class A(psd.AdaptiveDetector):
        def _score_from_record(self, timecode, record, height, width):
            return float(scores[timecode.frame_num])

    det = A(adaptive_threshold=athr, min_scene_len=length, window_width=w, min_content_val=mcv, engine=object())
    det.stats_manager = psd.StatsManager()
    want = []
    for i in range(n):
        want += [c.frame_num for c in det.process_record(FrameTimecode(i, fps), None, 1, 1)]
    got, ratio = epilogue.adaptive_cuts(scores, fps, athr, length, w, mcv)
    assert got == want, (seed, fps, length, w)
    key = det.get_metrics()[-1]
    for i in range(n):
        have = det.stats_manager.get_metrics(i, [key])[0]
        assert (have is None and np.isnan(ratio[i])) or have == ratio[i]


@pytest.mark.parametrize("seed", range(30))
def test_threshold_and_histogram(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3, 300))
    h, w = 6, 8
    fps = FPS[seed % len(FPS)]
    length = LENS[int(rng.integers(0, len(LENS)))]
    recs = np.zeros(n, RECORD_DTYPE)
    level = np.clip(np.cumsum(rng.normal(0, 12, n)) + 40, 0, 255)          # wanders through the threshold
    recs["byte_sum"] = (level * h * w * 3).astype(np.uint64)
    for t in range(n):
        recs["hist"][t] = np.bincount(rng.integers(0, 256, h * w) if rng.random() < 0.2 else
                                      np.clip(rng.normal(level[t], 10, h * w), 0, 255).astype(int), minlength=256)
    thr = int(rng.integers(5, 120))
    method = psd.ThresholdDetector.Method.FLOOR if seed % 2 else psd.ThresholdDetector.Method.CEILING
    bias = float(rng.choice([0.0, 0.5, -1.0, 1.0, 0.25]))
    final = bool(seed % 3 == 0)
    det = psd.ThresholdDetector(threshold=thr, min_scene_len=length, fade_bias=bias, add_final_scene=final, method=method,
                                engine=object())
    want = []
    for i in range(n):
        want += [c.frame_num for c in det.process_record(FrameTimecode(i, fps), recs[i], h, w)]
    want += [c.frame_num for c in det.post_process(FrameTimecode(n - 1, fps))]
    got, avg = epilogue.threshold_cuts(recs, h, w, fps, thr, length, bias, final, 0 if method == psd.ThresholdDetector.Method.FLOOR else 1)
    assert got == want, (seed, fps, length, method, bias, final)

    bins = int(rng.choice([256, 128, 64, 100, 17]))
    hthr = float(rng.uniform(0.02, 0.6))
    det = psd.HistogramDetector(threshold=hthr, bins=bins, min_scene_len=length, engine=object())
    det.stats_manager = psd.StatsManager()
    want = []
    for i in range(n):
        want += [c.frame_num for c in det.process_record(FrameTimecode(i, fps), recs[i], h, w)]
    got, diff = epilogue.hist_cuts(recs, fps, hthr, bins, length)
    assert got == want, (seed, fps, length, bins)
    key = det.get_metrics()[0]
    for i in range(1, n):
        assert det.stats_manager.get_metrics(i, [key])[0] == diff[i]


def test_per_frame_histogram_steps_native_equals_numpy():
    """HistogramDetector decides through psd_epilogue_hist_normalize / _correl; the numpy restatements next to them
    (cv2.calcHist + normalize + compareHist, operation for operation) must stay bit-identical."""
    import numpy as np

    from pyscenedetect_amd.detectors import histogram_detector as hd

    rng = np.random.default_rng(31)
    for bins in (128, 256, 100, 16, 7, 1):
        hists = [rng.integers(0, 5000, 256), np.zeros(256, np.int64), np.full(256, 8100), rng.integers(0, 3, 256) * 4_000_000,
                 np.eye(256, dtype=np.int64)[17] * 2_073_600]
        normed = []
        for h in hists:
            a, b = hd.normalized_histogram(h, bins), hd._native_normalized(h, bins)
            assert a.dtype == b.dtype == np.float32 and np.array_equal(a, b), bins
            normed.append(a)
        for x in normed:
            for y in normed:
                p, q = hd.compare_hist_correl(x, y), hd._native_correl(x, y)
                assert p == q or (p != p and q != q), (bins, p, q)


@pytest.mark.parametrize("seed", range(6))
def test_epilogues_read_sums_and_full_records_alike(seed):
    """ABI 3: ContentDetector / AdaptiveDetector / ThresholdDetector decide from the five sums of a record; the epilogues
    take them as full records (1064 B apart), as packed psd_frame_sums (40 B) or as a strided view, with one result."""
    from pyscenedetect_amd._native import SUMS_DTYPE

    rng = np.random.default_rng(300 + seed)
    n = int(rng.integers(1, 300))
    h, w = 36, 64
    full = np.zeros(n, RECORD_DTYPE)
    for name in ("sad_h", "sad_s", "sad_v", "edge_xor"):
        full[name] = rng.integers(0, 255 * h * w, n)
    full["byte_sum"] = rng.integers(0, 255 * 3 * h * w, n)
    full["hist"] = rng.integers(0, 1000, (n, 256))
    sums = np.empty(n, SUMS_DTYPE)
    for name in SUMS_DTYPE.names:
        sums[name] = full[name]
    every_other = np.repeat(sums, 2)[::2]          # a view with twice the stride
    weights = (1.0, 0.5, 2.0, 0.25)
    want = epilogue.content_scores(full, h, w, weights)
    for recs in (sums, every_other):
        got = epilogue.content_scores(recs, h, w, weights)
        assert all(np.array_equal(got[k], want[k]) for k in want)
    cuts, avg = epilogue.threshold_cuts(full, h, w, 25.0, threshold=100, min_scene_len=3)
    for recs in (sums, every_other):
        c2, a2 = epilogue.threshold_cuts(recs, h, w, 25.0, threshold=100, min_scene_len=3)
        assert c2 == cuts and np.array_equal(a2, avg)
    with pytest.raises(ValueError):
        epilogue.content_scores(np.zeros(4, np.uint64), h, w)
    with pytest.raises(ValueError):
        epilogue.hist_cuts(sums, 25.0)             # the histogram detector needs the histogram
