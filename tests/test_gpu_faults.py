"""GPU: device faults surface as Python exceptions carrying ``psd_last_error`` (errors from the device path must
come out of ``process_frame`` / ``detect_scenes`` like the reference's own exceptions, scene_manager.py:598-618), and the engine
stays usable afterwards.  Plus resource behaviour a long-running service depends on: the coefficient-table cache is bounded,
two engines share one GPU with their own edge workspaces."""
import threading

import numpy as np
import pytest

import pyscenedetect_amd as psd
from oracle.detectors_np import score_batch as oracle_score
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.synth import make_clip

pytestmark = pytest.mark.gpu


def test_allocation_beyond_hbm_raises_memory_error_and_engine_survives(hip_engine):
    with pytest.raises(MemoryError, match="hipMalloc"):
        hip_engine.alloc(1 << 42)                        # 4 TiB: no MI355X has that
    with pytest.raises((MemoryError, RuntimeError), match="hipHostMalloc|hipMalloc"):
        hip_engine.pinned_array((1 << 46,))              # 64 TiB of page-locked host memory
    frames, _ = make_clip(3, 6, 72, 128, shot_len=(2, 3))
    got = hip_engine.score_host(frames)
    want = oracle_score(frames)
    for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"):
        assert np.array_equal(got[f], want[f]), f


def test_detect_scenes_reports_a_device_allocation_failure(hip_engine):
    """A batch buffer that cannot be allocated (a million 1080p frames per batch: 6 TB) fails in the decode thread; the
    error leaves ``detect_scenes`` as MemoryError with the native message, after the thread has been stopped -- and the next
    call on the same engine works."""
    frames, _ = make_clip(5, 24, 1080, 1920, shot_len=(6, 9))
    sm = psd.SceneManager(engine=hip_engine, batch_frames=1_000_000)
    sm.auto_downscale = False
    sm.add_detector(psd.ContentDetector(engine=hip_engine))
    with pytest.raises(MemoryError, match="hipMalloc"):
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    assert threading.active_count() < 32
    ok = psd.SceneManager(engine=hip_engine, batch_frames=8)
    ok.auto_downscale = False
    ok.add_detector(psd.ContentDetector(min_scene_len=3, engine=hip_engine))
    assert ok.detect_scenes(psd.ArrayVideoStream(frames, 25.0)) == len(frames)
    assert ok.get_cut_list(show_warning=False)


def test_invalid_device_arguments_raise_value_error_with_the_native_message(hip_engine):
    buf = hip_engine.alloc(1 << 20)
    with pytest.raises(ValueError, match="invalid batch shape"):
        hip_engine.score_device(buf.ptr, 1, 0, 16)
    with pytest.raises(ValueError, match="kernel_size"):
        hip_engine.score_device(buf.ptr, 1, 16, 16, flags=E.SCORE_EDGES, edge_kernel=4)
    with pytest.raises(NotImplementedError, match="max 63"):
        hip_engine.score_device(buf.ptr, 1, 16, 16, flags=E.SCORE_EDGES, edge_kernel=65)
    with pytest.raises(ValueError, match="nothing submitted"):
        hip_engine.collect(1)
    buf.free()


def test_two_engines_share_one_gpu_with_their_own_edge_workspaces():
    """Two engines on device 0, each from its own thread, both with the edge term (each owns an edge workspace of up to 8 GiB
    and its own record slots): same records as the oracle, no interference."""
    frames, _ = make_clip(21, 40, 270, 480, shot_len=(5, 9), noise=3.0)
    frames[10:20, 60:140, 100:300] = (240, 30, 30)
    want = oracle_score(frames, edges=True)
    errors, results = [], {}

    def work(i):
        try:
            eng = E.ScoringEngine(0)
            for _ in range(3):
                results[i] = eng.score_host(frames, flags=E.SCORE_ALL)
            eng.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((i, ex))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist", "edge_xor"):
            assert np.array_equal(results[i][f], want[f]), (i, f)


def test_table_cache_is_bounded_and_evicted_shapes_come_back(hip_engine):
    """More distinct (source, target) shapes than the engine keeps tables for (48 per kind): the least recently used are
    freed, and a shape that was evicted is rebuilt with the same result."""
    import cv2  # the oracle shim

    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (1, 64, 96, 3), dtype=np.uint8)
    a = hip_engine.alloc(src.nbytes)
    a.upload(src.reshape(-1))
    b = hip_engine.alloc(64 * 96 * 3)
    first = None
    for rnd in range(2):
        for k in range(60):
            dh, dw = 8 + k % 30, 9 + k
            for inter in (cv2.INTER_LINEAR, cv2.INTER_NEAREST):
                hip_engine.resize_device(a.ptr, 1, 64, 96, b.ptr, dh, dw, interpolation=inter)
                got = b.download(dh * dw * 3).reshape(dh, dw, 3)
                assert np.array_equal(got, cv2.resize(src[0], (dw, dh), interpolation=inter)), (rnd, k, inter)
            thumbs = hip_engine.hash_thumbs_device(a.ptr, 1, 64, 96, 8 + k % 24)
            if first is None:
                first = thumbs.copy()
    assert np.array_equal(hip_engine.hash_thumbs_device(a.ptr, 1, 64, 96, 8), first)
    a.free()
    b.free()
