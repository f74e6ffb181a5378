"""If a REAL OpenCV build is importable (it is not in the build/bench image), pin the oracle to it.

The oracle is a restatement of OpenCV's published 8-bit algorithms ("parity unpinned at the cv2
boundary").  On any machine that does have ``cv2`` these tests close that gap:
every primitive of the hot path is compared with the real library on seeded inputs, and
``tools/dump_cv2_vectors.py`` can freeze the vectors into tests/golden/ for machines without it.
Skipped (not failed) when only the oracle shim is on the path.
"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")
if "oracle-shim" in getattr(cv2, "__version__", ""):
    pytest.skip("only the oracle's cv2 shim is importable: nothing to pin against", allow_module_level=True)

from oracle import lib as orc  # noqa: E402
from oracle.detectors_np import edge_map, hsv_planes  # noqa: E402


def _img(seed, h=97, w=131):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def test_hsv_and_yuv_all_triples():
    g8 = np.arange(256, dtype=np.uint8)
    for b0 in range(0, 256, 64):
        bb, gg, rr = np.meshgrid(g8[b0:b0 + 64], g8, g8, indexing="ij")
        img = np.ascontiguousarray(np.stack([bb, gg, rr], axis=-1).reshape(64 * 256, 256, 3))
        hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
        hp, sp, vp = hsv_planes(img)
        assert np.array_equal(hsv[..., 0], hp) and np.array_equal(hsv[..., 1], sp) and np.array_equal(hsv[..., 2], vp)
        y = np.empty(img.shape[:2], np.uint8)
        orc.lib().orc_bgr2y(img.ctypes.data, img.shape[1] * 3, y.ctypes.data, img.shape[0], img.shape[1])
        assert np.array_equal(cv2.cvtColor(img, cv2.COLOR_BGR2YUV)[..., 0], y)


@pytest.mark.parametrize("seed", range(4))
def test_canny_dilate_hist_resize(seed):
    img = _img(seed)
    lum = np.ascontiguousarray(cv2.cvtColor(img, cv2.COLOR_BGR2HSV)[..., 2])
    med = np.median(lum)
    low, high = int(max(0, (1 - 1 / 3) * med)), int(min(255, (1 + 1 / 3) * med))
    want = cv2.dilate(cv2.Canny(lum, low, high), np.ones((5, 5), np.uint8))
    assert np.array_equal(edge_map(img, 5), want)
    # histogram chain
    y = np.ascontiguousarray(cv2.cvtColor(img, cv2.COLOR_BGR2YUV)[..., 0])
    for bins in (256, 128, 100):
        h = cv2.calcHist([y], [0], None, [bins], [0, 256])
        n = cv2.normalize(h, h).flatten()
        ref = np.zeros((bins, 1), np.float32)
        orc.lib().orc_calc_hist_u8(y.ctypes.data, y.shape[1], y.shape[0], y.shape[1], bins, 0.0, 256.0, ref.ctypes.data)
        flat = ref.reshape(-1).copy()
        orc.lib().orc_normalize_l2_f32(flat.ctypes.data, bins)
        assert np.array_equal(n, flat)
        other = np.roll(flat, 3).copy()
        got = orc.lib().orc_compare_hist_correl(flat.ctypes.data, other.ctypes.data, bins)
        assert abs(got - cv2.compareHist(flat, other, cv2.HISTCMP_CORREL)) <= 4e-16   # lane order may differ by an ulp
    for dw, dh in ((64, 48), (65, 48), (131 // 2, 97 // 2)):
        out = np.empty((dh, dw, 3), np.uint8)
        orc.lib().orc_resize_linear_u8(img.ctypes.data, img.shape[1] * 3, img.shape[0], img.shape[1], 3, out.ctypes.data, dw * 3, dh, dw)
        assert np.array_equal(out, cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR))


@pytest.mark.parametrize("seed", range(4))
def test_hash_detector_primitives_and_other_resize_modes(seed):
    """BGR2GRAY, INTER_AREA (fractional, integer, 2x2), INTER_NEAREST bit-exact; cv2.dct within float32 noise
    (its operation order depends on the OpenCV build)."""
    img = _img(seed, 108, 192)
    gray = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
    mine = np.empty_like(gray)
    orc.lib().orc_bgr2gray(img.ctypes.data, img.shape[1] * 3, mine.ctypes.data, img.shape[1], img.shape[0], img.shape[1])
    assert np.array_equal(gray, mine)
    for size in (16, 32, 27, 54, 96):
        want = cv2.resize(gray, (size, size), interpolation=cv2.INTER_AREA)
        got = np.empty((size, size), np.uint8)
        assert orc.lib().orc_resize_area_u8(gray.ctypes.data, gray.shape[1], gray.shape[0], gray.shape[1], got.ctypes.data, size, size, size) == 0
        assert np.array_equal(got, want), size
    for dw, dh in ((96, 54), (64, 36), (100, 41), (192, 108)):
        for inter, fn in ((cv2.INTER_AREA, "orc_resize_area_u8_cn"), (cv2.INTER_NEAREST, "orc_resize_nearest_u8")):
            want = cv2.resize(img, (dw, dh), interpolation=inter)
            got = np.empty((dh, dw, 3), np.uint8)
            getattr(orc.lib(), fn)(img.ctypes.data, img.shape[1] * 3, img.shape[0], img.shape[1], 3, got.ctypes.data, dw * 3, dh, dw)
            assert np.array_equal(got, want), (dw, dh, fn)
    x = (np.float32(cv2.resize(gray, (32, 32), interpolation=cv2.INTER_AREA)) / 255).astype(np.float32)
    low = np.empty((16, 16), np.float32)
    orc.lib().orc_dct2d_f32(np.ascontiguousarray(x).ctypes.data, 32, 16, low.ctypes.data)
    assert np.abs(cv2.dct(x)[:16, :16] - low).max() <= 1e-5


@pytest.mark.parametrize("seed", range(3))
def test_lanczos4_and_cubic(seed):
    """INTER_LANCZOS4 has one 8-bit result in OpenCV (integer arithmetic) and must match; INTER_CUBIC has several: the
    installed build must be ONE of the three restated forms byte for byte -- or, on a build with IPP (x86-64 PyPI wheels; cv2.ipp.useIPP()),
    within one level of the `sse` form, which is all that can be said of a closed implementation.  The form found is printed: that is the
    value PSD_CUBIC_FORM should have next to this cv2."""
    img = np.random.default_rng(100 + seed).integers(0, 256, (360, 640, 3), dtype=np.uint8)
    img[:, :200] = (255, 0, 255)
    found = set(range(3))
    for dw, dh in ((256, 144), (255, 143), (700, 380)):
        got = np.empty((dh, dw, 3), np.uint8)
        orc.lib().orc_resize_lanczos4_u8(img.ctypes.data, 640 * 3, 360, 640, 3, got.ctypes.data, dw * 3, dh, dw)
        assert np.array_equal(got, cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LANCZOS4)), (dw, dh)
        want = cv2.resize(img, (dw, dh), interpolation=cv2.INTER_CUBIC)
        for form in range(3):
            orc.lib().orc_resize_cubic_u8(img.ctypes.data, 640 * 3, 360, 640, 3, got.ctypes.data, dw * 3, dh, dw, form)
            if not np.array_equal(got, want):
                found.discard(form)
            if form == 0:
                assert np.abs(got.astype(int) - want).max() <= 1, (dw, dh)
    ipp = bool(getattr(getattr(cv2, "ipp", None), "useIPP", lambda: False)())
    print("INTER_CUBIC of this cv2 is form(s)", sorted(found), "of (sse, fma, fixed); IPP in use:", ipp)
    assert found or ipp
