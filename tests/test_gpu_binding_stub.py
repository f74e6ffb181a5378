"""GPU box: the reference-side binding (integration/scenedetect_amd.py) gets the same records from libpsd_hip.so as from
the oracle-backed build of the same entry points -- the library the reference-side test (tests/test_reference_binding.py)
runs against in the build container, where the reference is available and this GPU is not."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_same_records_through_the_same_binding():
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import scenedetect_amd as B
    from pyscenedetect_amd.synth import make_clip

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libpsd_oracle_abi.so"])
    gpu = B.Binding(os.path.join(ROOT, "pyscenedetect_amd", "libpsd_hip.so"))
    cpu = B.Binding(os.path.join(ROOT, "oracle", "libpsd_oracle_abi.so"))
    frames, _ = make_clip(7, 8, 90, 160, shot_len=(3, 4), noise=4.0)
    frames[2, 20:50, 30:90] = (250, 20, 20)
    frames[3, 22:52, 34:94] = (250, 20, 20)
    for flags, kernel in ((B.HSV_SAD, 0), (B.HSV_SAD | B.EDGES, 0), (B.HSV_SAD | B.EDGES, 5), (B.LUMA_HIST, 0), (B.BYTE_SUM, 0),
                          (B.HSV_SAD | B.LUMA_HIST | B.BYTE_SUM | B.EDGES, 7)):
        for t in range(len(frames)):
            prev = frames[t - 1] if t else None
            a, b = gpu.score(frames[t], prev, flags, kernel), cpu.score(frames[t], prev, flags, kernel)
            assert bytes(a) == bytes(b), f"flags {flags} kernel {kernel} frame {t}"
    # ContentDetector's seam: the previous frame stays in device memory (FramePair), one upload per frame
    for flags, kernel in ((B.HSV_SAD, 0), (B.HSV_SAD | B.EDGES, 0), (B.HSV_SAD | B.EDGES, 5)):
        pg, pc = gpu.frame_pair(), cpu.frame_pair()
        for t in range(len(frames)):
            (a, ha), (b, hb) = pg.score_next(frames[t], flags, kernel), pc.score_next(frames[t], flags, kernel)
            assert ha == hb == (t > 0) and bytes(a) == bytes(b), f"pair: flags {flags} kernel {kernel} frame {t}"
            assert bytes(a) == bytes(gpu.score(frames[t], frames[t - 1] if t else None, flags, kernel))
        # a frame of another size is refused like the reference's own comparison refuses it (`assert left.shape == right.shape`,
        # content_detector.py:29-36) and the pair stays as it was; released, it starts over with any size
        other = np.ascontiguousarray(frames[0][:50, :70])
        with pytest.raises(AssertionError):
            pg.score_next(other, flags, kernel)
        assert bytes(pg.score_next(frames[-1], flags, kernel)[0]) == bytes(pc.score_next(frames[-1], flags, kernel)[0])
        pg.release()
        assert pg.score_next(other, flags, kernel)[1] is False and pg.score_next(other, flags, kernel)[1] is True
    # HashDetector's seam: the grey INTER_AREA thumbnail (psd_hash_thumbs)
    for size in (16, 32):
        for t in (0, 3, 7):
            assert np.array_equal(gpu.hash_thumb(frames[t], size), cpu.hash_thumb(frames[t], size)), f"thumb {size} frame {t}"
    with pytest.raises(ValueError):
        gpu.score(frames[0], None, B.EDGES, 4)
    with pytest.raises(ValueError):
        cpu.score(frames[0], None, B.EDGES, 4)
    gpu.close()
    cpu.close()
    assert ctypes.sizeof(B.FrameScores) == 1064


def test_batch_front_end_gets_the_same_records_from_both_libraries():
    """INTEGRATION.md B's batch front end (``Binding.score_videos_downscaled`` -> ``psd_score_segments_downscaled_device``): the golden
    corpus' videos of one size packed into one device batch behind the reference's default downscale -- byte-identical records from
    ``libpsd_hip.so`` and from the oracle-backed build, which is the library ``tests/test_reference_binding.py`` drives under the
    unmodified reference in the build container (there the decisions come out as the reference's own cut lists)."""
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    import scenedetect_amd as B

    from tests._helpers import corpus_clip

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libpsd_oracle_abi.so"])
    gpu = B.Binding(os.path.join(ROOT, "pyscenedetect_amd", "libpsd_hip.so"))
    cpu = B.Binding(os.path.join(ROOT, "oracle", "libpsd_oracle_abi.so"))
    for group, (dh, dw) in ((("bbc_a", "bbc_b", "bbc_c", "noisy_a"), (144, 256)), (("hd_b",), (144, 256)), (("odd_a",), (142, 256)),
                            (("small_a",), (120, 200))):
        videos = [corpus_clip(k) for k in group] + [np.zeros((0,) + corpus_clip(group[0]).shape[1:], np.uint8)]     # ... and one without frames
        for flags in (B.HSV_SAD, B.LUMA_HIST, B.BYTE_SUM, B.HSV_SAD | B.LUMA_HIST | B.BYTE_SUM):
            a, b = gpu.score_videos_downscaled(videos, dh, dw, flags), cpu.score_videos_downscaled(videos, dh, dw, flags)
            assert [len(x) for x in a] == [len(v) for v in videos]
            for k, (x, y) in enumerate(zip(a, b)):
                assert len(x) == len(y) and all(bytes(p) == bytes(q) for p, q in zip(x, y)), (group, flags, k)
    gpu.close()
    cpu.close()
