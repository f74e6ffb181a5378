"""CPU: the oracle (C restatement of the cv2 primitives) against independent numpy/scipy
formulations of the same published algorithms, and against itself on composition.  This is what
"pinned by construction" means while no real cv2 build is available."""
import numpy as np
import pytest
from scipy import ndimage

import cv2  # oracle/cv2_shim (tests/conftest.py puts it on sys.path)
from oracle import lib as orc
from oracle.detectors_np import canny_thresholds, edge_map, hsv_planes
from pyscenedetect_amd.detectors.histogram_detector import compare_hist_correl, normalized_histogram


def np_hsv(b, g, r):
    """Vectorised restatement of OpenCV's RGB2HSV_b (hsv_shift 12, hrange 180)."""
    b, g, r = (x.astype(np.int64) for x in (b, g, r))
    i = np.arange(256, dtype=np.float64)
    with np.errstate(divide="ignore"):
        sdiv = np.concatenate([[0], np.rint((255 << 12) / i[1:])]).astype(np.int64)
        hdiv = np.concatenate([[0], np.rint((180 << 12) / (6.0 * i[1:]))]).astype(np.int64)
    v = np.maximum(np.maximum(b, g), r)
    m = np.minimum(np.minimum(b, g), r)
    d = v - m
    s = (d * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * d, r - g + 4 * d))
    h = (h * hdiv[d] + 2048) >> 12          # numpy >> on negative int64 is arithmetic
    h = np.where(h < 0, h + 180, h)
    return h.astype(np.uint8), s.astype(np.uint8), v.astype(np.uint8)


def test_hsv_all_16777216_triples():
    g8 = np.arange(256, dtype=np.uint8)
    for b0 in range(0, 256, 32):                      # 8 slabs of 2M pixels
        bb, gg, rr = np.meshgrid(g8[b0:b0 + 32], g8, g8, indexing="ij")
        img = np.stack([bb, gg, rr], axis=-1).reshape(32 * 256, 256, 3)
        hp, sp, vp = hsv_planes(img)
        h, s, v = np_hsv(img[..., 0], img[..., 1], img[..., 2])
        assert np.array_equal(hp, h) and np.array_equal(sp, s) and np.array_equal(vp, v)
        assert hp.max() < 180
        # interleaved entry point agrees with the planar one
        inter = cv2.cvtColor(img[:256], cv2.COLOR_BGR2HSV)
        assert np.array_equal(inter[..., 0], hp[:256]) and np.array_equal(inter[..., 2], vp[:256])


def test_luma_and_yuv_formulas():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 97, 3), dtype=np.uint8)
    yuv = cv2.cvtColor(img, cv2.COLOR_BGR2YUV)
    b, g, r = (img[..., c].astype(np.int64) for c in range(3))
    y = (1868 * b + 9617 * g + 4899 * r + 8192) >> 14
    u = np.clip(((b - y) * 8061 + (128 << 14) + 8192) >> 14, 0, 255)
    v = np.clip(((r - y) * 14369 + (128 << 14) + 8192) >> 14, 0, 255)
    assert np.array_equal(yuv[..., 0], y) and np.array_equal(yuv[..., 1], u) and np.array_equal(yuv[..., 2], v)
    grey = np.full((4, 4, 3), 200, np.uint8)
    assert (cv2.cvtColor(grey, cv2.COLOR_BGR2YUV) == [200, 128, 128]).all()


@pytest.mark.parametrize("bins", [256, 128, 100, 7])
def test_hist_normalize_compare(bins):
    rng = np.random.default_rng(bins)
    planes = [rng.integers(0, 256, (50, 70), dtype=np.uint8), np.clip(rng.normal(90, 20, (50, 70)), 0, 255).astype(np.uint8)]
    hs = []
    for p in planes:
        hist = cv2.calcHist([p], [0], None, [bins], [0, 256])
        want = np.bincount(np.floor(p.reshape(-1).astype(np.float64) * bins / 256.0).astype(int), minlength=bins)
        assert hist.shape == (bins, 1) and np.array_equal(hist[:, 0], want.astype(np.float32))
        n = cv2.normalize(hist, hist).flatten()
        scale = np.float32(1.0 / np.sqrt(np.sum(want.astype(np.float64) ** 2)))
        assert np.array_equal(n, want.astype(np.float32) * scale)
        # the product's host epilogue computes the same thing from a 256-bin histogram
        assert np.array_equal(n, normalized_histogram(np.bincount(p.reshape(-1), minlength=256).astype(np.uint32), bins))
        hs.append(n)
    c = cv2.compareHist(hs[0], hs[1], cv2.HISTCMP_CORREL)
    assert c == compare_hist_correl(hs[0], hs[1])                       # bit-identical restatements
    assert abs(c - np.corrcoef(hs[0].astype(np.float64), hs[1].astype(np.float64))[0, 1]) < 1e-12
    assert cv2.compareHist(hs[0], hs[0], cv2.HISTCMP_CORREL) == pytest.approx(1.0, abs=1e-15)
    flat = np.ones(bins, np.float32)
    assert cv2.compareHist(flat, flat, cv2.HISTCMP_CORREL) == 1.0      # zero variance -> defined as 1


def np_canny(img, low, high):
    """Independent formulation: padded-array Sobel, vectorised NMS, hysteresis as connected components."""
    h, w = img.shape
    p = np.pad(img.astype(np.int64), 1, mode="edge")
    dx = (p[:-2, 2:] - p[:-2, :-2]) + 2 * (p[1:-1, 2:] - p[1:-1, :-2]) + (p[2:, 2:] - p[2:, :-2])
    dy = (p[2:, :-2] - p[:-2, :-2]) + 2 * (p[2:, 1:-1] - p[:-2, 1:-1]) + (p[2:, 2:] - p[:-2, 2:])
    mag = np.abs(dx) + np.abs(dy)
    mp = np.pad(mag, 1)                                   # zero magnitude outside the image
    c = mp[1:-1, 1:-1]
    ax, ay = np.abs(dx), np.abs(dy) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    s = np.where((dx ^ dy) < 0, -1, 1)
    yy, xx = np.mgrid[0:h, 0:w]
    up = mp[yy, xx + 1 - s]                               # mag[y-1][x-s] in padded coordinates
    dn = mp[yy + 2, xx + 1 + s]                           # mag[y+1][x+s]
    keep = np.where(horiz, (c > mp[1:-1, :-2]) & (c >= mp[1:-1, 2:]),
                    np.where(vert, (c > mp[:-2, 1:-1]) & (c >= mp[2:, 1:-1]), (c > up) & (c > dn)))
    cand = keep & (c > low)
    strong = cand & (c > high)
    lab, _ = ndimage.label(cand, structure=np.ones((3, 3), int))
    good = np.unique(lab[strong])
    return (np.isin(lab, good[good > 0]) * 255).astype(np.uint8)


@pytest.mark.parametrize("seed", range(6))
def test_canny_against_independent_formulation(seed):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(5, 70)), int(rng.integers(5, 90))
    if seed % 2:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:  # smooth structure + noise gives long weak chains for the hysteresis
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.clip(128 + 80 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    for low, high in ((20, 60), (0, 0), (100, 90), (300, 400)):
        got = cv2.Canny(img, low, high)
        lo, hi = min(low, high), max(low, high)
        assert np.array_equal(got, np_canny(img, lo, hi)), (seed, low, high)


@pytest.mark.parametrize("k", [3, 5, 13])
def test_dilate_against_scipy(k):
    rng = np.random.default_rng(k)
    img = (rng.random((40, 57)) > 0.97).astype(np.uint8) * 255
    got = cv2.dilate(img, np.ones((k, k), np.uint8))
    want = ndimage.maximum_filter(img, size=(k, k), mode="constant", cval=0)
    assert np.array_equal(got, want)


def test_edge_map_composition_and_thresholds():
    rng = np.random.default_rng(1)
    frame = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    _, _, lum = hsv_planes(frame)
    low, high = canny_thresholds(lum)
    med = float(np.median(lum))
    assert (low, high) == (int(max(0, (1.0 - 1.0 / 3.0) * med)), int(min(255, (1.0 + 1.0 / 3.0) * med)))
    assert np.array_equal(edge_map(frame, 5), cv2.dilate(cv2.Canny(lum, low, high), np.ones((5, 5), np.uint8)))


@pytest.mark.parametrize("shape", [((90, 160), (72, 128)), ((100, 300), (41, 77)), ((64, 64), (32, 32)), ((50, 50), (50, 50))])
def test_resize_properties(shape):
    (sh, sw), (dh, dw) = shape
    rng = np.random.default_rng(sh)
    img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    out = cv2.resize(img, (dw, dh))
    assert out.shape == (dh, dw, 3)
    const = np.full((sh, sw, 3), 137, np.uint8)
    assert (cv2.resize(const, (dw, dh)) == 137).all()
    if (sh, sw) == (dh, dw):
        assert np.array_equal(out, img)
    if (sh, sw) == (2 * dh, 2 * dw):
        s = img.astype(np.int32)
        assert np.array_equal(out, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    else:
        # float bilinear with OpenCV's half-pixel mapping, within fixed-point rounding (1 LSB)
        fy = np.clip((np.arange(dh) + 0.5) * sh / dh - 0.5, 0, sh - 1)
        fx = np.clip((np.arange(dw) + 0.5) * sw / dw - 0.5, 0, sw - 1)
        y0, x0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
        y1, x1 = np.minimum(y0 + 1, sh - 1), np.minimum(x0 + 1, sw - 1)
        wy, wx = (fy - y0)[:, None, None], (fx - x0)[None, :, None]
        f = img.astype(np.float64)
        ref = (f[y0][:, x0] * (1 - wy) * (1 - wx) + f[y0][:, x1] * (1 - wy) * wx + f[y1][:, x0] * wy * (1 - wx) + f[y1][:, x1] * wy * wx)
        assert np.abs(out.astype(np.float64) - ref).max() <= 1.0


def test_score_batch_is_the_composition_of_the_primitives():
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, (4, 30, 41, 3), dtype=np.uint8)
    rec = orc.score_batch(frames)
    for t in range(4):
        assert rec["byte_sum"][t] == frames[t].astype(np.int64).sum()
        y = cv2.cvtColor(frames[t], cv2.COLOR_BGR2YUV)[..., 0]
        assert np.array_equal(rec["hist"][t], np.bincount(y.reshape(-1), minlength=256))
        if t:
            for name, a, b in zip(("sad_h", "sad_s", "sad_v"), hsv_planes(frames[t]), hsv_planes(frames[t - 1])):
                assert rec[name][t] == np.abs(a.astype(np.int32) - b.astype(np.int32)).sum()
    assert rec["sad_h"][0] == 0


# ---- HashDetector primitives -----------------------------------------------------------------------

def test_gray_formula_and_extremes():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    img[0, 0] = (255, 255, 255)
    img[0, 1] = (0, 0, 0)
    L = orc.lib()
    g = np.empty((40, 50), np.uint8)
    L.orc_bgr2gray(img.ctypes.data, 150, g.ctypes.data, 50, 40, 50)
    b, gr, r = (img[..., c].astype(np.int64) for c in range(3))
    want = (b * 3735 + gr * 19235 + r * 9798 + (1 << 14)) >> 15
    assert np.array_equal(g, want)
    assert g[0, 0] == 255 and g[0, 1] == 0
    # the three weights are the 15-bit roundings of 0.114 / 0.587 / 0.299 and sum to one
    assert 3735 + 19235 + 9798 == 1 << 15


def _box_weights(ss, ds):
    sc = ss / ds
    m = np.zeros((ds, ss))
    for d in range(ds):
        a, b = d * sc, (d + 1) * sc
        for s in range(ss):
            m[d, s] = max(0.0, min(b, s + 1) - max(a, s)) / sc
    return m


@pytest.mark.parametrize("shape", [(72, 128, 16), (54, 96, 32), (37, 53, 16), (33, 47, 33), (108, 192, 16), (100, 100, 7)])
def test_area_resize_is_the_box_average(shape):
    """Fractional INTER_AREA == exact area-weighted mean up to float32 accumulation noise (then rounded)."""
    h, w, s = shape
    rng = np.random.default_rng(h + w)
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)
    out = np.empty((s, s), np.uint8)
    assert orc.lib().orc_resize_area_u8(g.ctypes.data, w, h, w, out.ctypes.data, s, s, s) == 0
    exact = _box_weights(h, s) @ g.astype(np.float64) @ _box_weights(w, s).T
    assert np.abs(out.astype(np.float64) - exact).max() <= 0.5 + 1e-3
    flat = np.full((h, w), 173, np.uint8)
    assert orc.lib().orc_resize_area_u8(flat.ctypes.data, w, h, w, out.ctypes.data, s, s, s) == 0
    assert (out == 173).all()


def test_area_resize_integer_scales():
    rng = np.random.default_rng(8)
    g = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    out = np.empty((32, 32), np.uint8)
    assert orc.lib().orc_resize_area_u8(g.ctypes.data, 64, 64, 64, out.ctypes.data, 32, 32, 32) == 0
    s = g.astype(np.int64)
    assert np.array_equal(out, (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2)
    g = rng.integers(0, 256, (96, 128), dtype=np.uint8)
    assert orc.lib().orc_resize_area_u8(g.ctypes.data, 128, 96, 128, out.ctypes.data, 32, 32, 32) == 0
    sums = g.astype(np.int64).reshape(32, 3, 32, 4).sum(axis=(1, 3))
    want = np.rint((sums.astype(np.float32) * np.float32(1.0 / 12)).astype(np.float64))  # half-to-even
    assert np.array_equal(out, want.astype(np.uint8))
    # not a decimation: refused
    assert orc.lib().orc_resize_area_u8(g.ctypes.data, 128, 96, 128, np.empty((200, 200), np.uint8).ctypes.data, 200, 200, 200) == -1


def test_dct_matches_scipy_and_hash_bits_are_balanced():
    import scipy.fft

    rng = np.random.default_rng(3)
    x = rng.random((32, 32)).astype(np.float32)
    low = np.empty((16, 16), np.float32)
    orc.lib().orc_dct2d_f32(x.ctypes.data, 32, 16, low.ctypes.data)
    ref = scipy.fft.dctn(x.astype(np.float64), norm="ortho")[:16, :16]
    assert np.abs(low - ref).max() <= 4e-6
    thumbs = rng.integers(0, 256, (5, 32, 32), dtype=np.uint8)
    thumbs[4] = 0  # all-black frame: max_value guard (hash_detector.py:132-135)
    bits = orc.hash_bits(thumbs, 16)
    assert bits.shape == (5, 16, 16)
    assert all(b.sum() == 128 for b in bits[:4])  # strictly-above-median of 256 distinct values
    assert bits[4].sum() == 0


def test_nearest_and_multichannel_area_resize():
    import cv2  # the oracle shim

    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    for (dh, dw) in ((41, 77), (97, 131), (10, 13), (48, 65)):
        got = cv2.resize(img, (dw, dh), interpolation=cv2.INTER_NEAREST)
        ys = np.minimum(np.floor(np.arange(dh) * (1.0 / (dh / 97))).astype(int), 96)
        xs = np.minimum(np.floor(np.arange(dw) * (1.0 / (dw / 131))).astype(int), 130)
        assert np.array_equal(got, img[ys][:, xs])
        # INTER_AREA treats the interleaved channels independently
        area = cv2.resize(img, (dw, dh), interpolation=cv2.INTER_AREA)
        for c in range(3):
            assert np.array_equal(area[..., c], cv2.resize(np.ascontiguousarray(img[..., c]), (dw, dh), interpolation=cv2.INTER_AREA))
    assert cv2.resize(img, (200, 200), interpolation=cv2.INTER_AREA).shape == (200, 200, 3)   # enlargement: bilinear emulation
    with pytest.raises(NotImplementedError):
        cv2.resize(img, (50, 50), interpolation=7)


def _cubic_numpy(img, dh, dw, form):
    """cv2.resize(INTER_CUBIC) for 8-bit images written a second time, in numpy: resize.cpp's HResizeCubic in int32, then the vertical pass in
    the form asked for -- float32 arrays (numpy rounds every product and every sum: the unfused form; the fused one through float64, in
    which a float32 product is exact and one rounding is left) or FixedPtCast."""
    def taps(ssize, dsize):
        d = np.arange(dsize)
        f = ((d + 0.5) * (1.0 / (dsize / ssize)) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        x = (f - s.astype(np.float32)).astype(np.float32)
        A, one = np.float32(-0.75), np.float32(1)
        c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
        c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
        c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
        c3 = one - c0 - c1 - c2
        c = np.stack([c0, c1, c2, c3], 1)
        assert c.dtype == np.float32
        return s - 1, np.rint(c * np.float32(2048)).astype(np.int64)
    sh, sw = img.shape[:2]
    xf, xa = taps(sw, dw)
    yf, yb = taps(sh, dh)
    cols = np.clip(xf[:, None] + np.arange(4), 0, sw - 1)                     # (dw, 4)
    hor = (img.astype(np.int64)[:, cols, :] * xa[None, :, :, None]).sum(2)    # (sh, dw, 3)
    rows = np.clip(yf[:, None] + np.arange(4), 0, sh - 1)                     # (dh, 4)
    S = hor[rows].reshape(dh, 4, dw * 3)                                      # (dh, 4, width)
    fixed = np.clip(((S * yb[:, :, None]).sum(1) + (1 << 21)) >> 22, 0, 255)
    if form == 2:
        return fixed.astype(np.uint8).reshape(dh, dw, 3)
    b = (yb.astype(np.float32) * np.float32(1.0 / (2048 * 2048)))[:, :, None]
    Sf = S.astype(np.float32)
    if form == 0:
        t = Sf[:, 3] * b[:, 3]
        for k in (2, 1, 0):
            t = Sf[:, k] * b[:, k] + t
            assert t.dtype == np.float32
    else:
        t = (Sf[:, 3] * b[:, 3]).astype(np.float64)
        for k in (2, 1, 0):     # (a float32 product is exact in float64: what is rounded to float32 is product + t, as fmaf does)
            t = (Sf[:, k].astype(np.float64) * b[:, k].astype(np.float64) + t).astype(np.float32).astype(np.float64)
    vec = np.clip(np.rint(t), 0, 255)
    width = dw * 3
    out = np.where(np.arange(width)[None, :] < width - width % 8, vec, fixed)
    return out.astype(np.uint8).reshape(dh, dw, 3)


@pytest.mark.parametrize("shape", [((97, 131), (41, 77)), ((360, 640), (144, 256)), ((540, 960), (143, 255)), ((9, 5), (3, 2)), ((4, 4), (31, 17)),
                                   ((50, 70), (50, 70)), ((45, 80), (90, 161))])
def test_cubic_resize_forms_against_a_second_restatement(shape):
    """orc_resize_cubic_u8 (C) against the numpy restatement above in all three forms, on noise with saturated halves; what the forms have
    in common (identity at equal sizes, constants stay constant, never more than one level apart)."""
    from oracle import lib as orc

    (sh, sw), (dh, dw) = shape
    rng = np.random.default_rng(sh * 5 + dw)
    img = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    img[:, : max(1, sw // 3)] = (255, 0, 255)
    outs = []
    for form in range(3):
        got = np.empty((dh, dw, 3), np.uint8)
        orc.lib().orc_resize_cubic_u8(img.ctypes.data, sw * 3, sh, sw, 3, got.ctypes.data, dw * 3, dh, dw, form)
        want = _cubic_numpy(img, dh, dw, form)
        assert np.array_equal(got, want), (form, int(np.count_nonzero(got != want)))
        outs.append(got.astype(int))
    assert max(np.abs(outs[0] - outs[1]).max(), np.abs(outs[0] - outs[2]).max()) <= 1
    if dh * dw > 1000:
        assert outs[0].min() == 0 and outs[0].max() == 255
    if (sh, sw) == (dh, dw):
        assert all(np.array_equal(o, img) for o in outs)
    flat = np.full((sh, sw, 3), 201, np.uint8)
    got = np.empty((dh, dw, 3), np.uint8)
    orc.lib().orc_resize_cubic_u8(flat.ctypes.data, sw * 3, sh, sw, 3, got.ctypes.data, dw * 3, dh, dw, 0)
    assert np.all(np.abs(got.astype(int) - 201) <= 1)       # (the 11-bit weights of a tap set need not sum to 2048 exactly)


def test_cubic_form_switch_of_the_shim(monkeypatch):
    import cv2  # the oracle shim
    from oracle import lib as orc

    img = np.random.default_rng(3).integers(0, 256, (360, 640, 3), dtype=np.uint8)
    for name, form in cv2.CUBIC_FORMS.items():
        monkeypatch.setenv("PSD_CUBIC_FORM", name)
        want = np.empty((143, 255, 3), np.uint8)
        orc.lib().orc_resize_cubic_u8(img.ctypes.data, 640 * 3, 360, 640, 3, want.ctypes.data, 255 * 3, 143, 255, form)
        assert np.array_equal(cv2.resize(img, (255, 143), interpolation=cv2.INTER_CUBIC), want)
    monkeypatch.delenv("PSD_CUBIC_FORM")
    monkeypatch.setenv("PSD_CUBIC_FORM", "ipp")
    with pytest.raises(KeyError):
        cv2.resize(img, (255, 143), interpolation=cv2.INTER_CUBIC)


def test_fp32_formulation_of_the_hsv_pass_is_exact_on_all_triples():
    """The float32 arithmetic the HSV-only kernel uses (psd_score_kernels.hip, pixel_fp_*), replayed on the host over all
    2^24 BGR triples against the oracle (tools/hsv_fp32_check.py); the -m gpu twin runs the kernel itself."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "hsv_fp32_check.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all 2^24 triples: 0" in out.stdout


def test_inter_area_upscaling_is_bilinear_with_area_coefficients():
    """cv2.resize(INTER_AREA) that does not shrink along both axes: OpenCV's emulation (resize.cpp) -- for an exact 2x
    enlargement every source pixel is simply repeated (fx = (dx + 1) - (sx + 1) * 2 <= 0 -> weight 0 on the right tap), a
    constant image stays constant, and sizes that do shrink go through the true area path as before."""
    import cv2

    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    up = cv2.resize(img, (28, 18), interpolation=cv2.INTER_AREA)
    assert np.array_equal(up, np.repeat(np.repeat(img, 2, axis=0), 2, axis=1))
    flat = np.full((7, 5, 3), 93, np.uint8)
    assert (cv2.resize(flat, (13, 9), interpolation=cv2.INTER_AREA) == 93).all()
    # 1.5x: destination pixel 1 of a row straddles source pixels 0 and 1 (sx = 0, fx = 2 - 1 * 1.5 = 0.5)
    row = np.array([[[0, 0, 0], [200, 100, 50]]], np.uint8)
    got = cv2.resize(row, (3, 1), interpolation=cv2.INTER_AREA)[0, :, 0].tolist()
    assert got == [0, 100, 200]
    # mixed: wider but lower -- both axes take the emulation
    mixed = cv2.resize(img, (20, 6), interpolation=cv2.INTER_AREA)
    assert mixed.shape == (6, 20, 3)
    down = cv2.resize(img, (7, 3), interpolation=cv2.INTER_AREA)      # shrinks along both: the area path
    assert down.shape == (3, 7, 3)
