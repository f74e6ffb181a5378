"""ctypes loader for oracle/liboracle.so (C restatement of the cv2 primitives).

TEST INFRASTRUCTURE ONLY (see oracle/cv2_restate.c header).  PARITY UNPINNED at the cv2 boundary:
no real OpenCV is available in this environment.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc if missing or stale."""
    src = os.path.join(_HERE, "cv2_restate.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _SO


class ScoreRecord(ctypes.Structure):
    """Mirror of psd_frame_scores (include/psd_engine.h)."""

    _fields_ = [
        ("sad_h", ctypes.c_uint64),
        ("sad_s", ctypes.c_uint64),
        ("sad_v", ctypes.c_uint64),
        ("edge_xor", ctypes.c_uint64),
        ("byte_sum", ctypes.c_uint64),
        ("hist", ctypes.c_uint32 * 256),
    ]


RECORD_DTYPE = np.dtype(
    [
        ("sad_h", "<u8"),
        ("sad_s", "<u8"),
        ("sad_v", "<u8"),
        ("edge_xor", "<u8"),
        ("byte_sum", "<u8"),
        ("hist", "<u4", (256,)),
    ]
)
assert RECORD_DTYPE.itemsize == ctypes.sizeof(ScoreRecord) == 1064

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, sz, i, d = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_double
        L.orc_hsv_tables.argtypes = [vp, vp]
        L.orc_bgr2hsv.argtypes = [vp, sz, vp, sz, i, i]
        L.orc_bgr2hsv_planes.argtypes = [vp, sz, vp, vp, vp, i, i]
        L.orc_bgr2yuv.argtypes = [vp, sz, vp, sz, i, i]
        L.orc_bgr2y.argtypes = [vp, sz, vp, i, i]
        L.orc_calc_hist_u8.argtypes = [vp, sz, i, i, i, d, d, vp]
        L.orc_normalize_l2_f32.argtypes = [vp, i]
        L.orc_compare_hist_correl.argtypes = [vp, vp, i]
        L.orc_compare_hist_correl.restype = d
        L.orc_sobel3.argtypes = [vp, sz, i, i, vp, vp]
        L.orc_canny.argtypes = [vp, sz, i, i, d, d, vp]
        L.orc_dilate_rect.argtypes = [vp, sz, i, i, i, i, vp]
        L.orc_resize_linear_u8.argtypes = [vp, sz, i, i, i, vp, sz, i, i]
        L.orc_resize_area_upscale_u8.restype = None
        L.orc_resize_area_upscale_u8.argtypes = [vp, sz, i, i, i, vp, sz, i, i]
        L.orc_bgr2gray.argtypes = [vp, sz, vp, sz, i, i]
        L.orc_resize_area_u8.argtypes = [vp, sz, i, i, vp, sz, i, i]
        L.orc_resize_area_u8_cn.argtypes = [vp, sz, i, i, i, vp, sz, i, i]
        L.orc_resize_nearest_u8.argtypes = [vp, sz, i, i, i, vp, sz, i, i]
        L.orc_resize_lanczos4_u8.restype = None
        L.orc_resize_lanczos4_u8.argtypes = [vp, sz, i, i, i, vp, sz, i, i]
        L.orc_lanczos4_taps.restype = None
        L.orc_lanczos4_taps.argtypes = [i, i, vp, vp]
        L.orc_resize_cubic_u8.restype = None
        L.orc_resize_cubic_u8.argtypes = [vp, sz, i, i, i, vp, sz, i, i, i]
        L.orc_cubic_taps.restype = None
        L.orc_cubic_taps.argtypes = [i, i, vp, vp]
        L.orc_dct2d_f32.argtypes = [vp, i, i, vp]
        L.orc_score_batch.argtypes = [vp, i, i, i, sz, sz, vp, vp]
        L.orc_score_batch_flags.argtypes = [vp, i, i, i, sz, sz, vp, vp, ctypes.c_uint]
        _lib = L
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def hsv_tables():
    s = np.zeros(256, np.int32)
    h = np.zeros(256, np.int32)
    lib().orc_hsv_tables(_p(s), _p(h))
    return s, h


def score_batch(frames: np.ndarray, prev: np.ndarray | None = None, flags: int = 7) -> np.ndarray:
    """Oracle per-frame integer records for ``frames`` uint8[N,H,W,3] (BGR).

    Returns a structured array (RECORD_DTYPE).  ``edge_xor`` is left 0 here; see
    :func:`oracle.detectors_np.edge_maps` for the edge term.
    """
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    assert frames.ndim == 4 and frames.shape[3] == 3
    n, h, w, _ = frames.shape
    out = np.zeros(n, RECORD_DTYPE)
    pp = None
    if prev is not None:
        prev = np.ascontiguousarray(prev, dtype=np.uint8)
        assert prev.shape == (h, w, 3)
        pp = _p(prev)
    if n:
        lib().orc_score_batch_flags(_p(frames), n, h, w, w * 3, h * w * 3, pp, _p(out), int(flags) & 7)
    return out


def hash_thumbs(frames: np.ndarray, size: int) -> np.ndarray:
    """Grey ``size x size`` INTER_AREA thumbnails of BGR frames [n,h,w,3] -- the integer/float32 front half
    of ``HashDetector.hash_frame`` (reference ``hash_detector.py:125-129``)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, h, w, _ = frames.shape
    L = lib()
    gray = np.empty((h, w), np.uint8)
    out = np.empty((n, size, size), np.uint8)
    for t in range(n):
        L.orc_bgr2gray(_p(frames[t]), w * 3, _p(gray), w, h, w)
        if size > w or size > h:
            # not shrunk along both axes: OpenCV's bilinear emulation with area-mode coefficients (resize.cpp), as in the shim
            L.orc_resize_area_upscale_u8(_p(gray), w, h, w, 1, _p(out[t]), size, size, size)
        elif L.orc_resize_area_u8(_p(gray), w, h, w, _p(out[t]), size, size, size) != 0:
            raise NotImplementedError("INTER_AREA restated for decimation only")
    return out


def hash_bits(thumbs: np.ndarray, hash_size: int) -> np.ndarray:
    """Back half of ``hash_frame`` (``hash_detector.py:131-151``): scale by the maximum, DCT, keep the low
    ``hash_size`` frequencies, threshold at their median.  Returns bool [n, hash_size, hash_size]."""
    thumbs = np.ascontiguousarray(thumbs, dtype=np.uint8)
    n, s, _ = thumbs.shape
    L = lib()
    bits = np.empty((n, hash_size, hash_size), bool)
    low = np.empty((hash_size, hash_size), np.float32)
    for t in range(n):
        mx = thumbs[t].max()
        if mx == 0:
            mx = 1
        x = np.ascontiguousarray(np.float32(thumbs[t]) / mx)
        L.orc_dct2d_f32(_p(x), s, hash_size, _p(low))
        bits[t] = low > np.median(low)
    return bits
