"""Deterministic synthetic clips shaped like real footage (numpy only).

Used by the tests, the golden-fixture generator and ``bench.py``: shots built from a smooth random
base image plus per-frame noise, separated by hard cuts, with some shots fading to black and back
(distribution "S").  The same seed gives the same bytes wherever numpy's version is
the same, which holds between the build container and the GPU box.
"""

import numpy as np


def _smooth_image(rng: np.random.Generator, h: int, w: int, grid=(9, 16)) -> np.ndarray:
    gh, gw = grid
    g = rng.integers(0, 256, (gh, gw, 3)).astype(np.float64)
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.minimum(ys.astype(int), gh - 2) if gh > 1 else np.zeros(h, int)
    x0 = np.minimum(xs.astype(int), gw - 2) if gw > 1 else np.zeros(w, int)
    wy = (ys - y0)[:, None, None]
    wx = (xs - x0)[None, :, None]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - wy) * (1 - wx) + b * (1 - wy) * wx + c * wy * (1 - wx) + d * wy * wx)


def make_clip(seed: int, n_frames: int, height: int, width: int, shot_len=(12, 40), fade_every: int = 3,
              noise: float = 2.0, fade_len: int = 8):
    """Returns ``(frames uint8[N,H,W,3], cuts)`` where ``cuts`` are the first frames of new shots."""
    rng = np.random.default_rng(seed)
    frames = np.empty((n_frames, height, width, 3), np.uint8)
    cuts = []
    t = 0
    shot = 0
    while t < n_frames:
        length = int(rng.integers(shot_len[0], shot_len[1] + 1))
        base = _smooth_image(rng, height, width)
        drift = rng.normal(0, 0.15, 3)
        fade = fade_every > 0 and shot % fade_every == fade_every - 1 and length > 2 * fade_len + 2
        if shot > 0:
            cuts.append(t)
        for k in range(length):
            if t >= n_frames:
                break
            img = base + drift * k + rng.normal(0, noise, base.shape)
            if fade:
                gain = min(1.0, k / fade_len, (length - 1 - k) / fade_len)
                img = img * max(0.0, gain)
            frames[t] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
            t += 1
        shot += 1
    return frames, cuts


def make_clip_fast(seed: int, n_frames: int, height: int, width: int, shot_len=(16, 40), fade_every: int = 3, fade_len: int = 6,
                   noise: int = 5):
    """Like :func:`make_clip` but in integer arithmetic, for fixtures at 1080p / 4K (50 instead of 2000 ms per 1080p frame):
    shots of a smooth random image with three flat rectangles on it (sharp edges for a resize to interpolate), one of them
    moving, per-frame uniform noise of 0..``noise - 1``, hard cuts, every ``fade_every``-th long shot fading in and out.
    Returns ``(frames uint8[N,H,W,3], cuts)``; the same seed gives the same bytes wherever numpy's version is the same."""
    rng = np.random.default_rng(seed)
    frames = np.empty((n_frames, height, width, 3), np.uint8)
    cuts, t, shot = [], 0, 0
    wide = np.empty((height, width, 3), np.uint16)
    ys = np.linspace(0, 8, height, dtype=np.float32)
    xs = np.linspace(0, 15, width, dtype=np.float32)
    y0, x0 = np.minimum(ys.astype(np.int32), 7), np.minimum(xs.astype(np.int32), 14)
    wy, wx = (ys - y0).astype(np.float32)[:, None, None], (xs - x0).astype(np.float32)[None, :, None]
    while t < n_frames:
        length = int(rng.integers(shot_len[0], shot_len[1] + 1))
        g = rng.integers(8, 244, (9, 16, 3)).astype(np.float32)
        cols = g[:, x0] * (1 - wx) + g[:, x0 + 1] * wx                      # (9, width, 3): separable, columns first
        base = np.rint(cols[y0] * (1 - wy) + cols[y0 + 1] * wy).astype(np.uint8)          # in [8, 243]
        if noise > 12:
            np.minimum(base, 255 - noise, out=base)                        # (heavy noise: keep base + noise inside a byte)
        rects = []
        for _ in range(3):
            rh, rw = max(1, int(rng.integers(height // 12, height // 3 + 1))), max(1, int(rng.integers(width // 12, width // 3 + 1)))
            ry, rx = int(rng.integers(0, max(1, height - rh))), int(rng.integers(0, max(1, width - rw)))
            rects.append((ry, rx, rh, rw, rng.integers(8, 244 if noise <= 12 else 256 - noise, 3).astype(np.uint8)))
        for ry, rx, rh, rw, col in rects[:2]:
            base[ry:ry + rh, rx:rx + rw] = col
        step = int(rng.integers(1, max(2, width // 160)))
        fade = fade_every > 0 and shot % fade_every == fade_every - 1 and length > 2 * fade_len + 2
        if shot > 0:
            cuts.append(t)
        for k in range(length):
            if t >= n_frames:
                break
            img = base + rng.integers(0, noise, base.shape, dtype=np.uint8)
            ry, rx, rh, rw, col = rects[2]
            xk = (rx + k * step) % max(1, width - rw)
            img[ry:ry + rh, xk:xk + rw] = col
            gain = min(1.0, k / fade_len, (length - 1 - k) / fade_len) if fade else 1.0
            if gain < 1.0:
                # (img * gain256) >> 8 through one reused 16-bit buffer: fresh 12 MB temporaries cost 0.1 - 0.5 s each in page faults
                np.multiply(img, np.uint16(int(max(0.0, gain) * 256)), out=wide, dtype=np.uint16)
                np.right_shift(wide, 8, out=wide)
                frames[t] = wide
            else:
                frames[t] = img
            t += 1
        shot += 1
    return frames, cuts
