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
