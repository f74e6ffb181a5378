"""Exhaustive check (all 2^24 BGR triples, numpy float32) of the fp32-pipe formulation of OpenCV's 8-bit HSV that
DESIGN.md section 7 lists as the next candidate for the HSV pass: B, G, R as floats, max / min / differences / the
2*diff and 4*diff terms as float adds and fused multiply-adds, S = trunc(fma(diff, sdiv[v] * 2^-12, 0.5)); only the
hue product (more than 24 bits) stays integer.  Every float intermediate is an integer (or an integer / 4096) below
2^24, so float32 holds it exactly -- this script confirms it against the oracle's tables and formulas.
    python tools/hsv_fp32_check.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lib as orc  # noqa: E402

sdiv, hdiv = orc.hsv_tables()
sdiv_f = (sdiv.astype(np.float64) / 4096.0).astype(np.float32)
assert np.array_equal(sdiv_f.astype(np.float64) * 4096.0, sdiv.astype(np.float64)), "sdiv * 2^-12 must be exact in float32"
g8 = np.arange(256, dtype=np.uint8)
bad = 0
for b0 in range(0, 256, 32):
    bb, gg, rr = np.meshgrid(g8[b0:b0 + 32], g8, g8, indexing="ij")
    img = np.ascontiguousarray(np.stack([bb, gg, rr], axis=-1).reshape(32 * 256, 256, 3))
    want = np.empty_like(img)
    orc.lib().orc_bgr2hsv(img.ctypes.data, 256 * 3, want.ctypes.data, 256 * 3, img.shape[0], 256)
    b, g, r = (img[..., c].astype(np.float32) for c in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vi, di = v.astype(np.int64), diff.astype(np.int64)
    s = np.trunc(diff * sdiv_f[vi] + np.float32(0.5))            # product and sum are exact, so fma == mul + add
    assert s.dtype == np.float32
    c_r = g - b
    c_g = (b - r) + np.float32(2.0) * diff
    c_b = (r - g) + np.float32(4.0) * diff
    hraw = np.where(v == r, c_r, np.where(v == g, c_g, c_b)).astype(np.int64)
    hh = (hraw * hdiv[di].astype(np.int64) + 2048) >> 12
    hh = np.where(hh < 0, hh + 180, hh)
    got = np.stack([hh, s.astype(np.int64), vi], axis=-1)
    bad += int(np.count_nonzero(got != want.astype(np.int64)))
print("fp32-pipe HSV formulation: mismatching channel values over all 2^24 triples:", bad)
sys.exit(1 if bad else 0)
