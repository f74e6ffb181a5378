"""pytest plugin for tests/test_reference_own_tests.py: what the reference's conftest would provide to its test_api.py workflows --
a `test_video_file` fixture (here the name of the alias package's synthetic clip) -- and, on a box without a GPU, the CPU oracle engine
as the mirror's default engine (test infrastructure: the product has no CPU path and raises without a GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture
def test_video_file(tmp_path):
    return str(tmp_path / "synthetic_clip.mp4")


def pytest_configure(config):
    sys.path[:0] = [p for p in (os.path.join(ROOT, "oracle", "cv2_shim"), ROOT) if p not in sys.path]
    from pyscenedetect_amd import engine

    try:
        have_gpu = engine.device_count() > 0
    except Exception:  # noqa: BLE001 -- no library, no runtime: no GPU
        have_gpu = False
    if not have_gpu:
        from oracle.detectors_np import OracleEngine

        shared = OracleEngine()
        engine.default_engine = lambda device=None: shared
