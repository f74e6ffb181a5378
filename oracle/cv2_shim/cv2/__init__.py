"""Minimal ``cv2`` stand-in -- TEST INFRASTRUCTURE ONLY.

Put ``oracle/cv2_shim`` and ``/root/reference`` on ``sys.path`` and the *unmodified* reference
package (``scenedetect``) imports and runs its detectors on top of the C restatement in
``oracle/cv2_restate.c``.  Used only in the build container to validate the oracle and to generate
the golden fixtures under ``tests/golden/`` (``oracle/gen_golden.py``); it is never imported by the
product and never runs on the GPU box as part of the product path.

PARITY UNPINNED: this is a restatement of OpenCV's published 8-bit algorithms, not OpenCV.
Only the functions the hot path calls are provided.
"""

import os

import numpy as np

from oracle import lib as _orc

__version__ = "0.0-oracle-shim"

# Constants with OpenCV's numeric values.
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_CUBIC = 2
INTER_AREA = 3
INTER_LANCZOS4 = 4
CUBIC_FORMS = {"sse": 0, "fma": 1, "fixed": 2}      # (the shim's own: which OpenCV build's INTER_CUBIC is restated)
COLOR_BGR2RGB = 4
COLOR_BGR2GRAY = 6
COLOR_BGR2HSV = 40
COLOR_BGR2YUV = 82
HISTCMP_CORREL = 0
NORM_L2 = 4
CAP_PROP_POS_MSEC = 0
CAP_PROP_POS_FRAMES = 1
CAP_PROP_FRAME_WIDTH = 3
CAP_PROP_FRAME_HEIGHT = 4
CAP_PROP_FPS = 5
CAP_PROP_FOURCC = 6
CAP_PROP_FRAME_COUNT = 7
CAP_PROP_SAR_NUM = 40
CAP_PROP_SAR_DEN = 41
CAP_PROP_ORIENTATION_AUTO = 49
IMWRITE_JPEG_QUALITY = 1
IMWRITE_PNG_COMPRESSION = 16


class VideoCapture:  # only referenced as a type annotation / isinstance target
    def __init__(self, *a, **k):
        raise NotImplementedError("cv2 shim has no video decoding")


def _u8c3(img):
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("cv2 shim: expected uint8 HxWx3")
    return np.ascontiguousarray(img)


def cvtColor(src, code):
    src = _u8c3(src)
    h, w, _ = src.shape
    dst = np.empty_like(src)
    L = _orc.lib()
    if code == COLOR_BGR2HSV:
        L.orc_bgr2hsv(src.ctypes.data, w * 3, dst.ctypes.data, w * 3, h, w)
    elif code == COLOR_BGR2YUV:
        L.orc_bgr2yuv(src.ctypes.data, w * 3, dst.ctypes.data, w * 3, h, w)
    elif code == COLOR_BGR2GRAY:
        dst = np.empty((h, w), np.uint8)
        L.orc_bgr2gray(src.ctypes.data, w * 3, dst.ctypes.data, w, h, w)
    else:
        raise NotImplementedError(f"cv2 shim: cvtColor code {code}")
    return dst


def split(m):
    m = np.asarray(m)
    return tuple(np.ascontiguousarray(m[:, :, c]) for c in range(m.shape[2]))


def Canny(image, threshold1, threshold2):
    image = np.ascontiguousarray(image, dtype=np.uint8)
    assert image.ndim == 2
    h, w = image.shape
    dst = np.empty((h, w), np.uint8)
    _orc.lib().orc_canny(image.ctypes.data, w, h, w, float(threshold1), float(threshold2), dst.ctypes.data)
    return dst


def dilate(src, kernel):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    kernel = np.asarray(kernel)
    assert src.ndim == 2 and kernel.ndim == 2 and bool(np.all(kernel != 0)), "rect kernels only"
    h, w = src.shape
    dst = np.empty((h, w), np.uint8)
    _orc.lib().orc_dilate_rect(src.ctypes.data, w, h, w, kernel.shape[0], kernel.shape[1], dst.ctypes.data)
    return dst


def calcHist(images, channels, mask, histSize, ranges):
    assert len(images) == 1 and list(channels) == [0] and mask is None and len(histSize) == 1
    plane = np.ascontiguousarray(images[0], dtype=np.uint8)
    assert plane.ndim == 2
    h, w = plane.shape
    bins = int(histSize[0])
    hist = np.zeros((bins, 1), np.float32)
    _orc.lib().orc_calc_hist_u8(plane.ctypes.data, w, h, w, bins, float(ranges[0]), float(ranges[1]), hist.ctypes.data)
    return hist


def normalize(src, dst=None):
    out = np.array(src, dtype=np.float32, copy=True)
    flat = out.reshape(-1)
    _orc.lib().orc_normalize_l2_f32(flat.ctypes.data, flat.size)
    return out


def compareHist(H1, H2, method):
    assert method == HISTCMP_CORREL
    a = np.ascontiguousarray(H1, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(H2, dtype=np.float32).reshape(-1)
    assert a.size == b.size
    return float(_orc.lib().orc_compare_hist_correl(a.ctypes.data, b.ctypes.data, a.size))


def resize(src, dsize, interpolation=INTER_LINEAR):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dw, dh = int(dsize[0]), int(dsize[1])
    dst = np.empty((dh, dw) + (() if src.ndim == 2 else (cn,)), np.uint8)
    if interpolation == INTER_AREA:
        if dw > sw or dh > sh:
            # not shrunk along both axes: OpenCV emulates INTER_AREA with bilinear passes and area-mode coefficients
            _orc.lib().orc_resize_area_upscale_u8(src.ctypes.data, sw * cn, sh, sw, cn, dst.ctypes.data, dw * cn, dh, dw)
        elif _orc.lib().orc_resize_area_u8_cn(src.ctypes.data, sw * cn, sh, sw, cn, dst.ctypes.data, dw * cn, dh, dw) != 0:
            raise NotImplementedError("cv2 shim: INTER_AREA failed")
        return dst
    if interpolation == INTER_NEAREST:
        _orc.lib().orc_resize_nearest_u8(src.ctypes.data, sw * cn, sh, sw, cn, dst.ctypes.data, dw * cn, dh, dw)
        return dst
    if interpolation == INTER_LANCZOS4:
        _orc.lib().orc_resize_lanczos4_u8(src.ctypes.data, sw * cn, sh, sw, cn, dst.ctypes.data, dw * cn, dh, dw)
        return dst
    if interpolation == INTER_CUBIC:
        # the 8-bit vertical pass of INTER_CUBIC depends on the OpenCV build (cv2_restate.c: orc_resize_cubic_u8); PSD_CUBIC_FORM names the
        # one restated, as it does for the device library: sse (default) | fma | fixed
        form = CUBIC_FORMS[os.environ.get("PSD_CUBIC_FORM", "sse")]
        _orc.lib().orc_resize_cubic_u8(src.ctypes.data, sw * cn, sh, sw, cn, dst.ctypes.data, dw * cn, dh, dw, form)
        return dst
    if interpolation != INTER_LINEAR:
        raise NotImplementedError("cv2 shim: INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA and INTER_LANCZOS4 only")
    dst = np.empty((dh, dw) + (() if src.ndim == 2 else (cn,)), np.uint8)
    _orc.lib().orc_resize_linear_u8(src.ctypes.data, sw * cn, sh, sw, cn, dst.ctypes.data, dw * cn, dh, dw)
    return dst


def dct(src):
    src = np.ascontiguousarray(src, dtype=np.float32)
    n = src.shape[0]
    if src.ndim != 2 or src.shape[1] != n:
        raise NotImplementedError("cv2 shim: dct of square float32 blocks only")
    dst = np.empty((n, n), np.float32)
    _orc.lib().orc_dct2d_f32(src.ctypes.data, n, n, dst.ctypes.data)
    return dst
