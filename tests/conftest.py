import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The cv2 shim is only needed where the oracle restates cv2.resize for the downscale path.
SHIM = os.path.join(ROOT, "oracle", "cv2_shim")
if SHIM not in sys.path:
    sys.path.append(SHIM)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def _gpu_available() -> bool:
    try:
        from pyscenedetect_amd import engine

        return engine.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    # `pytest -m gpu` asks for the GPU tests explicitly: without a usable GPU / libpsd_hip.so they must FAIL loudly
    # (engine creation raises), not pass by being skipped.  Only an unfiltered run on a CPU box skips them.
    markexpr = (config.getoption("markexpr", "") or "").strip()
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    skip = pytest.mark.skip(reason="no AMD GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_runs.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_engine():
    from oracle.detectors_np import OracleEngine

    return OracleEngine()


@pytest.fixture(scope="session")
def hip_engine():
    from pyscenedetect_amd.engine import ScoringEngine

    eng = ScoringEngine(0)
    yield eng
    eng.close()


_clip_cache = {}


def golden_clip(golden, name):
    """Regenerate the frames of a golden clip from its seed (and check they are the same bytes)."""
    if name in _clip_cache:
        return _clip_cache[name]
    from pyscenedetect_amd.synth import make_clip

    c = golden["clips"][name]
    if c.get("uniform"):
        frames = np.random.default_rng(c["seed"]).integers(0, 256, (c["n"], c["h"], c["w"], 3), dtype=np.uint8)
    else:
        kw = dict(c["kwargs"])
        if "shot_len" in kw:
            kw["shot_len"] = tuple(kw["shot_len"])
        frames, _ = make_clip(c["seed"], c["n"], c["h"], c["w"], **kw)
    assert int(frames.sum()) == c["sum_all"], "synthetic clip differs from the one the golden run used"
    _clip_cache[name] = frames
    return frames
