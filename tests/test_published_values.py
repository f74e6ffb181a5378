"""CPU: the oracle against the handful of values OpenCV ITSELF publishes for the primitives on the path -- its documentation and
tutorials -- which are the only upstream-held numbers that exist without a cv2 build (the cv2 boundary stays
"parity unpinned"; these pin the restatement at the points upstream prints).

* ``py_colorspaces`` tutorial ("How to find HSV values to track?"): ``cv.cvtColor(np.uint8([[[0,255,0]]]), cv.COLOR_BGR2HSV)``
  prints ``[[[ 60 255 255]]]``; the same page's ranges: hue in [0,179], saturation and value in [0,255].
* ``imgproc`` colour-conversion documentation, RGB <-> HSV for 8-bit images: ``V = max(R,G,B)``, ``S = (V - min) / V`` (0 where
  V = 0), ``H = 60 (G - B) / (V - min)`` if V = R, ``120 + 60 (B - R) / (V - min)`` if V = G, ``240 + 60 (R - G) / (V - min)`` if
  V = B, ``H += 360 if H < 0``; 8-bit output ``V <- 255 V, S <- 255 S, H <- H / 2``.  At the six primary and secondary colours,
  white, black and the greys these formulas have exact integer values -- no rounding rule is involved.
* the same documentation, RGB -> GRAY: ``Y = 0.299 R + 0.587 G + 0.114 B`` (also the Y of YUV / YCrCb): 76 / 150 / 29 for pure red,
  green and blue at 8 bits, 255 for white.
* ``cv.normalize`` with its defaults (``NORM_L2``, ``alpha = 1``) leaves a vector of Euclidean norm 1; ``cv.compareHist(h, h,
  HISTCMP_CORREL)`` is 1 and the correlation of a histogram with its mirror image about the mean is -1 (the documented formula
  ``sum((a - mean a)(b - mean b)) / sqrt(sum((a - mean a)^2) sum((b - mean b)^2))``).
* ``cv.dilate`` with a k x k rectangle turns one set pixel into a k x k block (morphology tutorial); ``cv.Canny`` of a constant
  image has no edges, and a step whose Sobel magnitude exceeds the upper threshold gives a one-pixel-wide line along the step."""
import numpy as np
import pytest

import cv2  # oracle/cv2_shim (tests/conftest.py puts it on sys.path)

PUBLISHED_HSV = {
    # BGR -> H (degrees / 2), S, V.  The green row is printed in the tutorial; the others follow from the documented formulas exactly.
    (0, 255, 0): (60, 255, 255),        # green
    (255, 0, 0): (120, 255, 255),       # blue
    (0, 0, 255): (0, 255, 255),         # red
    (0, 255, 255): (30, 255, 255),      # yellow
    (255, 255, 0): (90, 255, 255),      # cyan
    (255, 0, 255): (150, 255, 255),     # magenta
    (255, 255, 255): (0, 0, 255),       # white
    (0, 0, 0): (0, 0, 0),               # black
    (128, 128, 128): (0, 0, 128),       # grey
    (0, 128, 0): (60, 255, 128),        # dark green: S and H do not depend on the brightness of a pure colour
    (0, 0, 51): (0, 255, 51),
    (255, 255, 51): (90, 204, 255),     # S = 255 (255 - 51) / 255 = 204, H = 120 + 60 (255 - 51) / 204 = 180 degrees
}


@pytest.mark.parametrize("bgr,hsv", sorted(PUBLISHED_HSV.items()))
def test_hsv_at_the_points_the_documentation_fixes(bgr, hsv):
    px = np.array([[bgr]], np.uint8)
    assert tuple(int(x) for x in cv2.cvtColor(px, cv2.COLOR_BGR2HSV)[0, 0]) == hsv


def test_hue_range_is_0_to_179_and_value_is_the_maximum():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
    assert hsv[..., 0].max() <= 179
    assert np.array_equal(hsv[..., 2], img.max(axis=2))
    assert np.all(hsv[..., 1][img.max(axis=2) == img.min(axis=2)] == 0)


@pytest.mark.parametrize("bgr,y", [((0, 0, 255), 76), ((0, 255, 0), 150), ((255, 0, 0), 29), ((255, 255, 255), 255), ((0, 0, 0), 0)])
def test_grey_and_luma_of_the_primaries(bgr, y):
    px = np.array([[bgr]], np.uint8)
    assert int(cv2.cvtColor(px, cv2.COLOR_BGR2GRAY)[0, 0]) == y
    assert int(cv2.cvtColor(px, cv2.COLOR_BGR2YUV)[0, 0, 0]) == y


def test_normalize_defaults_and_correlation_extremes():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (48, 48), dtype=np.uint8)
    h = cv2.calcHist([img], [0], None, [128], [0, 256])
    assert h.shape == (128, 1) and h.dtype == np.float32 and int(h.sum()) == img.size
    n = cv2.normalize(h, h.copy())
    assert abs(float(np.sqrt((n.astype(np.float64) ** 2).sum())) - 1.0) < 1e-6
    assert cv2.compareHist(n, n, cv2.HISTCMP_CORREL) == pytest.approx(1.0, abs=1e-12)
    a = np.arange(128, dtype=np.float32).reshape(128, 1)
    assert cv2.compareHist(a, a[::-1].copy(), cv2.HISTCMP_CORREL) == pytest.approx(-1.0, abs=1e-12)
    flat = np.ones((128, 1), np.float32)
    assert cv2.compareHist(flat, flat, cv2.HISTCMP_CORREL) == 1.0      # zero variance: OpenCV returns 1


@pytest.mark.parametrize("k", [3, 5, 13])
def test_dilate_grows_one_pixel_into_a_block(k):
    img = np.zeros((41, 41), np.uint8)
    img[20, 20] = 255
    out = cv2.dilate(img, np.ones((k, k), np.uint8))
    want = np.zeros_like(img)
    want[20 - k // 2:20 + k // 2 + 1, 20 - k // 2:20 + k // 2 + 1] = 255
    assert np.array_equal(out, want)


def test_canny_of_flat_and_step_images():
    assert not cv2.Canny(np.full((32, 32), 90, np.uint8), 50, 150).any()
    step = np.zeros((32, 32), np.uint8)
    step[:, 16:] = 200                      # Sobel |dx| = 800 along the step, far above the upper threshold
    e = cv2.Canny(step, 50, 150)
    assert set(np.unique(e)) == {0, 255}
    cols = np.flatnonzero(e.any(axis=0))
    assert len(cols) == 1 and cols[0] in (15, 16)
    assert e[:, cols[0]].all()
