"""Exhaustive host replay (all 2^24 BGR triples, numpy float32 / float64) of the fp32-pipe formulation of OpenCV's 8-bit
HSV that the HSV-only scoring pass runs (pixel_fp_* in pyscenedetect_amd/csrc/psd_score_kernels.hip):

  channel   -> float 2^23 + byte (the kernel builds it with one v_perm_b32; the bias cancels in every difference)
  v, vmin   -> max / min of the biased floats; diff = v - vmin (exact)
  S         -> low byte of RNE(diff * sdiv' + 2^23), sdiv' = nextafter(sdiv / 4096): the one-ulp bump sends the exact .5
               ties of diff * sdiv / 4096 upwards like OpenCV's (x + 2048) >> 12 and moves nothing else
  v==r, v==g-> the 0/1 floats nr = clamp(v - r), ng = clamp(v - g); hraw = p + nr * ((q - p + 2 diff) + ng * (2 diff - p - 2 q))
               with p = g - b, q = b - r: nine full-rate instructions
  H         -> t = fma(hraw, hdiv / 4096, 2^-13) (exact), low 16 bits of RNE(t + 1.5 * 2^23) = floor(x + .5) as a two's
               complement, then min(h, h + 180) on unsigned 16-bit values

  luma      -> (fused all-detectors pass, pixel_fp_luma_bits) the BGR -> Y coefficients sum to 2^14, so
               y = g + floor((-6767 p - 4899 q) / 16384 + 1/2); u = fma(p, -6767/16384, fma(q, -4899/16384, 2^-15)) is exact
               and the low byte of RNE((2^23 + g) + u) is y

Every fused multiply-add is evaluated in float64 and rounded once to float32 (exact products: 8 x 24 and 11 x 17 bits),
which is what the hardware fma does.  The -m gpu test test_exhaustive_hsv_through_the_kernel runs the same triples
through the kernel itself.
    python tools/hsv_fp32_check.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lib as orc  # noqa: E402

f32, f64 = np.float32, np.float64
sdiv, hdiv = orc.hsv_tables()
sdivf = (sdiv.astype(f64) / 4096.0).astype(f32)
assert np.array_equal(sdivf.astype(f64) * 4096.0, sdiv.astype(f64)), "sdiv / 4096 must be exact in float32"
sdivf = np.where(sdiv != 0, np.nextafter(sdivf, f32(np.inf)), f32(0))
hdivf = (hdiv.astype(f64) / 4096.0).astype(f32)
assert np.array_equal(hdivf.astype(f64) * 4096.0, hdiv.astype(f64)), "hdiv / 4096 must be exact in float32"


def fma(a, b, c):
    """float32 fused multiply-add: exact in float64 for these operand widths, one rounding to float32."""
    return (a.astype(f64) * b.astype(f64) + f64(c) if np.isscalar(c) else a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


g8 = np.arange(256, dtype=np.uint8)
bad = 0
bad_y = 0
BIAS = f32(8388608.0)
for b0 in range(0, 256, 32):
    bb, gg, rr = np.meshgrid(g8[b0:b0 + 32], g8, g8, indexing="ij")
    img = np.ascontiguousarray(np.stack([bb, gg, rr], axis=-1).reshape(32 * 256, 256, 3))
    want = np.empty_like(img)
    orc.lib().orc_bgr2hsv(img.ctypes.data, 256 * 3, want.ctypes.data, 256 * 3, img.shape[0], 256)
    B, G, R = (img[..., c].astype(f32) + BIAS for c in range(3))
    V = np.maximum(np.maximum(B, G), R)
    diff = V - np.minimum(np.minimum(B, G), R)
    vi, di = (V - BIAS).astype(np.int64), diff.astype(np.int64)
    s_bits = fma(diff, sdivf[vi], BIAS).view(np.uint32) & 0xFF
    p, q = G - B, B - R
    w = fma(diff, np.full_like(diff, 2.0), -p)
    a, b = w + q, fma(q, np.full_like(q, -2.0), w)
    nm_r, nm_g = np.clip(V - R, 0, 1).astype(f32), np.clip(V - G, 0, 1).astype(f32)
    hraw = fma(nm_r, fma(nm_g, b, a), p)
    t = fma(hraw, hdivf[di], f32(2.0 ** -13))
    hb = (t + f32(12582912.0)).view(np.uint32) & 0xFFFF
    h = np.minimum(hb, (hb + 180) & 0xFFFF)
    got = np.stack([h, s_bits, vi], axis=-1)
    bad += int(np.count_nonzero(got != want.astype(np.int64)))
    u = fma(p, np.full_like(p, -6767.0 / 16384.0), fma(q, np.full_like(q, -4899.0 / 16384.0), f32(2.0 ** -15)))
    y_bits = (G + u).view(np.uint32)
    i64 = img.astype(np.int64)
    want_y = (1868 * i64[..., 0] + 9617 * i64[..., 1] + 4899 * i64[..., 2] + 8192) >> 14   # oracle/cv2_restate.c: B2Y, G2Y, R2Y, shift 14
    bad_y += int(np.count_nonzero((y_bits & 0xFF) != want_y)) + int(np.count_nonzero((y_bits >> 8) != (0x4B000000 >> 8)))
print("fp32-pipe HSV formulation: mismatching channel values over all 2^24 triples:", bad)
print("fp32-pipe luma from the hue differences: mismatching values over all 2^24 triples:", bad_y)
sys.exit(1 if bad or bad_y else 0)
