"""CPU: the C-ABI library loads and exports every symbol include/psd_engine.h declares; argument
validation and the no-GPU failure mode are loud (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from pyscenedetect_amd import _native, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "psd_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psd_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _native.load()
    names = _declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in psd_engine.h but not exported by libpsd_hip.so"
        assert name in _native.SYMBOLS, f"{name} has no ctypes binding"
    assert set(_native.SYMBOLS) == set(names)


def test_record_layout_matches_header():
    assert _native.RECORD_DTYPE.itemsize == 1064
    assert _native.RECORD_DTYPE.fields["hist"][1] == 40
    assert _native.load().psd_abi_version() == _native.ABI_VERSION == 8


def test_hsv_tables_match_oracle():
    from oracle import lib as orc

    s, h = engine.hsv_tables()
    so, ho = orc.hsv_tables()
    assert np.array_equal(s, so) and np.array_equal(h, ho)
    # the values the fixed-point formulas are defined by
    assert s[1] == 255 << 12 and s[255] == 4096 and h[1] == 122880 and h[255] == 482


def test_invalid_arguments_raise_value_error():
    lib = _native.load()
    with pytest.raises(ValueError):
        _native.check(lib.psd_epilogue_content_cuts(None, 0, 0, 25, 1, None, None, None))
    p = _native.AdaptiveParams()
    p.window_width = 0
    cuts = np.zeros(4, np.int64)
    nc = ctypes.c_int()
    cv = np.zeros(3)
    with pytest.raises(ValueError):
        _native.check(lib.psd_epilogue_adaptive_cuts(cv.ctypes.data, 3, 0, 25, 1, ctypes.byref(p), None,
                                                     cuts.ctypes.data, ctypes.byref(nc)))
    assert "invalid" in _native.last_error()


def test_no_gpu_fails_loudly():
    if engine.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.ScoringEngine(0)
    import pyscenedetect_amd as psd

    det = psd.ContentDetector()
    with pytest.raises(RuntimeError):
        det.process_frame(psd.FrameTimecode(0, 25.0), np.zeros((8, 8, 3), np.uint8))


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "pyscenedetect_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_missing_library_fails_loudly(tmp_path):
    """No HIP library -> an exception at first use, never a silent CPU path."""
    import subprocess
    import sys

    code = ("import numpy as np, pyscenedetect_amd as psd\n"
            "try:\n"
            "    psd.ContentDetector().process_frame(psd.FrameTimecode(0, 25.0), np.zeros((8, 8, 3), np.uint8))\n"
            "except Exception as ex:\n"
            "    print(type(ex).__name__, '|', ex)\n"
            "else:\n"
            "    print('NO ERROR')\n")
    env = dict(os.environ, PSD_LIB_PATH=str(tmp_path / "libpsd_missing.so"), PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.stdout.startswith("NativeLibraryError"), out.stdout + out.stderr
    assert "no CPU fallback" in out.stdout


def test_default_engine_is_per_thread_and_never_closed_from_another_thread(monkeypatch):
    """Host logic of `engine.default_engine` without a GPU (the engine class is replaced by a stand-in)."""
    import threading

    from pyscenedetect_amd import engine as E

    made = []

    class Fake:
        def __init__(self, device):
            self.device, self._h, self.closed = device, object(), False
            made.append(self)

        def close(self):
            self._h, self.closed = None, True

    monkeypatch.setattr(E, "ScoringEngine", Fake)
    monkeypatch.setattr(E, "device_count", lambda: 2)
    monkeypatch.setattr(E, "_default_tls", threading.local())
    a = E.default_engine(0)
    assert E.default_engine(0) is a and E.default_engine(1) is not a
    held = {}
    t = threading.Thread(target=lambda: held.setdefault("e", E.default_engine(0)))
    t.start()
    t.join()
    assert held["e"] is not a
    t2 = threading.Thread(target=lambda: E.default_engine(0))     # used to close the engines of dead threads
    t2.start()
    t2.join()
    assert not held["e"].closed and held["e"]._h is not None
    a.close()
    assert E.default_engine(0) is not a                            # a closed engine is replaced, not handed out
