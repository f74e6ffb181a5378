"""INTEGRATION.md B, executed: the UNMODIFIED reference package, its seams patched by integration/scenedetect_amd.py,
driven by its own SceneManager, reproduces the golden runs.

Where this runs: the build container (the reference lives at /root/reference there and nowhere else); without a GPU the
binding loads oracle/libpsd_oracle_abi.so -- the same five C-ABI entry points on the CPU oracle (test infrastructure).
With a GPU visible (and the reference present) the very same test loads pyscenedetect_amd/libpsd_hip.so instead.
tests/test_gpu_binding_stub.py covers the GPU box, where the reference is absent: identical records from both
libraries through the identical binding.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "scenedetect")),
                                reason="the reference checkout is only present in the build container")


def _lib_path():
    from tests.conftest import _gpu_available

    if os.environ.get("PSD_ORACLE_ABI_LIB"):          # (tools/sanitize/run_oracle_sanitized.py: the CPU build under ASan / UBSan)
        return os.environ["PSD_ORACLE_ABI_LIB"]
    if _gpu_available():
        return os.path.join(ROOT, "pyscenedetect_amd", "libpsd_hip.so")
    import subprocess

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libpsd_oracle_abi.so"])
    return os.path.join(ROOT, "oracle", "libpsd_oracle_abi.so")


@pytest.fixture(scope="module")
def patched_reference():
    shim = os.path.join(ROOT, "oracle", "cv2_shim")
    added = [p for p in (shim, REFERENCE) if p not in sys.path]
    sys.path[:0] = added
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import scenedetect  # noqa: F401  (the reference, unmodified)
    import scenedetect_amd

    assert os.path.realpath(scenedetect.__file__).startswith(REFERENCE)
    binding = scenedetect_amd.Binding(_lib_path())
    undo = scenedetect_amd.install(binding)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_golden  # the reference-side MemoryStream / run() that produced the goldens

    gen_golden.amd = (scenedetect_amd, binding)      # (for the tests that drive the binding's own entry points)
    yield gen_golden
    undo()
    binding.close()


CASES = [(clip, cfg) for clip in ("scenes_a", "fades_b", "ragged_c", "wide_d", "uniform_u")
         for cfg in ("content_default", "content_stats", "content_edges", "content_luma_suppress", "content_kernel5_secs",
                     "adaptive_default", "adaptive_w3", "hist_default", "hist_256", "hist_100", "threshold_default",
                     "threshold_final", "threshold_ceiling")]


@pytest.mark.parametrize("clip,cfg", CASES)
def test_reference_with_bound_seams_reproduces_the_goldens(patched_reference, golden, clip, cfg):
    from tests._helpers import assert_same_run
    from tests.conftest import golden_clip

    want = golden["clips"][clip]["results"].get(cfg)
    if want is None:
        pytest.skip("not part of this clip's golden set")
    cls_name, kwargs, with_stats = golden["configs"][cfg]
    frames = golden_clip(golden, clip)
    got = patched_reference.run(frames, cls_name, kwargs, with_stats, clip == "wide_d")
    assert_same_run(got, want, f"{clip}/{cfg}")


def test_binding_only_touches_the_c_abi():
    src = open(os.path.join(ROOT, "integration", "scenedetect_amd.py")).read()
    assert "import pyscenedetect_amd" not in src and "from pyscenedetect_amd" not in src
    assert "import ctypes" in src and "psd_score_batch" in src


def test_bound_content_detector_refuses_a_size_change_like_the_plain_one(patched_reference):
    """Per-frame API: ``ContentDetector.process_frame`` on a frame of another size than its predecessor raises AssertionError in the
    plain reference (``_mean_pixel_distance``, content_detector.py:29-36) and keeps the planes it had; the bound seam used to start
    over with a score of 0.0 (``--binding --wide`` campaign of round 5, seed 109 case 60 and 17 more)."""
    import scenedetect

    frames = np.random.default_rng(3).integers(0, 256, (6, 40, 64, 3), dtype=np.uint8)
    det = scenedetect.detectors.ContentDetector()
    base = scenedetect.FrameTimecode(0, 25.0)
    for i in range(3):
        det.process_frame(base + i, frames[i])
    with pytest.raises(AssertionError):
        det.process_frame(base + 3, np.ascontiguousarray(frames[3][:30, :50]))
    stats = scenedetect.StatsManager()          # the frame after is scored against frame 2, the last one that was accepted
    det.stats_manager = stats
    det.process_frame(base + 4, frames[4])
    want = np.abs(frames[4].max(axis=2).astype(int) - frames[2].max(axis=2).astype(int)).mean()
    assert stats.get_metrics(4, ["delta_lum"])[0] == want


def test_bound_hash_detector_takes_its_thumbnails_from_the_c_abi(patched_reference, golden):
    """``HashDetector.hash_frame``'s ``cvtColor(BGR2GRAY)`` + ``resize(INTER_AREA)`` lines bound to ``psd_hash_thumbs``
    (hash_detector.py:125-129; the DCT / median lines stay): same cuts and ``hash_dist`` metrics as the unbound reference, for
    thumbnails smaller and larger than the frame."""
    import scenedetect
    import scenedetect_amd
    from scenedetect.detectors import HashDetector

    from tests.conftest import golden_clip

    frames = golden_clip(golden, "scenes_a")
    calls = []
    original = scenedetect_amd.Binding.hash_thumb

    def spy(self, frame, size):
        calls.append(size)
        return original(self, frame, size)

    def run(**kw):
        stats = scenedetect.StatsManager()
        sm = scenedetect.SceneManager(stats)
        sm.auto_downscale = False
        det = HashDetector(**kw)
        sm.add_detector(det)
        sm.detect_scenes(patched_reference.MemoryStream(frames, 25.0))
        key = det.get_metrics()[0]
        return ([c.frame_num for c in sm.get_cut_list(show_warning=False)],
                [stats.get_metrics(i, [key])[0] if stats.metrics_exist(i, [key]) else None for i in range(len(frames))])

    for kw in ({}, {"size": 8, "lowpass": 4}, {"size": 32, "lowpass": 3}):          # 32, 32 and 96-pixel thumbnails of 72 x 128 frames
        scenedetect_amd.Binding.hash_thumb = spy
        try:
            bound = run(**kw)
        finally:
            scenedetect_amd.Binding.hash_thumb = original
        n_calls = len(calls)
        assert n_calls >= len(frames)
        saved = HashDetector.hash_frame
        # the unbound method: what install() saved
        import runpy

        import scenedetect.detectors.hash_detector as module

        plain_class = runpy.run_path(module.__file__)["HashDetector"]      # (a scratch copy of the source: nothing is reloaded)
        try:
            HashDetector.hash_frame = staticmethod(plain_class.hash_frame)
            plain = run(**kw)
        finally:
            HashDetector.hash_frame = staticmethod(saved)
        assert len(calls) == n_calls                       # the plain run did not touch the binding
        assert bound == plain and len(bound[0]) >= 2, kw
        calls.clear()


@pytest.mark.parametrize("group", [("bbc_a", "bbc_b", "bbc_c", "noisy_a"), ("qhd_a",), ("odd_a",), ("portrait_a",), ("small_a",), ("edge_a",)])
def test_batch_front_end_over_the_c_abi_gives_the_references_cut_lists(patched_reference, group):
    """INTEGRATION.md B's batch front end, executed: ``scenedetect_amd.detect_many`` packs decoded videos of one size into one device
    batch, takes their records from ONE ``psd_score_segments_downscaled_device`` call (every frame resized as the reference's
    SceneManager does by default) and lets the UNMODIFIED reference's own detector objects decide from them -- and returns the cut
    lists ``detect(video, detector_cls())`` returned for every video (``tests/golden/corpus_default_pipeline.json``, generated by the
    unpatched reference), for all four detectors: 640 x 360 (four videos in one batch), 960 x 540, 486 x 270, portrait, a size that is
    not resized at all and one that is resized by 0.4 %."""
    from scenedetect.detectors import AdaptiveDetector, ContentDetector, HistogramDetector, ThresholdDetector

    from tests._helpers import corpus_clip, corpus_golden

    amd, binding = patched_reference.amd
    videos = [corpus_clip(k) for k in group]
    makers = {"content": ContentDetector, "adaptive": lambda: AdaptiveDetector(window_width=2, min_content_val=15.0),
              "hist": HistogramDetector, "threshold": ThresholdDetector}
    for key, make in makers.items():
        got = amd.detect_many(binding, videos, make, fps=25.0)
        for name, cuts in zip(group, got):
            assert cuts == corpus_golden()["clips"][name]["cuts"][key], (key, name)
    assert binding.current is None
    # ... and the per-frame seams still work afterwards (the replay mode is off again)
    import scenedetect

    sm = scenedetect.SceneManager()
    sm.add_detector(ContentDetector())
    sm.detect_scenes(patched_reference.MemoryStream(videos[0], 25.0))
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == corpus_golden()["clips"][group[0]]["cuts"]["content"]
