"""Synthetic shot-like clips generated where they are scored (torch tensors on the engine's device; CPU tensors in the
dry-run tests): the inputs of BASELINE.json configs 4 and 5 in `bench.py --workload bbc|corpus`, `tools/bbc_standin.py`
and the GPU flow tests.  Bench / test infrastructure only -- nothing in `pyscenedetect_amd/` imports this.

A clip is a sequence of shots of random length; a shot is a smooth random image (bilinearly upsampled 16 x 9 grid) with
a slow colour drift and per-frame N(0, 2) noise; hard cuts between shots; every fifth shot (when long enough) fades in
from and out to black over 20 frames (distribution S).  The generator returns the ground-truth cut list.
"""
import numpy as np
import torch


def make_device_clip(seed: int, n: int, h: int, w: int, device, shot_len=(40, 400), block_bytes: int = 1 << 30, out=None):
    """(uint8[n, h, w, 3] on `device` -- `out` if given --, [first frame of every shot but the first])."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rs = np.random.default_rng(seed)
    x = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device) if out is None else out
    per = max(1, block_bytes // (h * w * 3 * 4))     # frames per float32 block
    cuts, t, shot = [], 0, 0
    while t < n:
        length = int(rs.integers(shot_len[0], shot_len[1]))
        grid = torch.rand((1, 3, 9, 16), device=device, generator=g) * 255.0
        base = torch.nn.functional.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
        drift = torch.randn((3,), device=device, generator=g) * 0.05
        fade = shot % 5 == 4 and length > 60
        if shot:
            cuts.append(t)
        m = min(length, n - t)
        for a in range(0, m, per):
            b = min(m, a + per)
            k = torch.arange(a, b, device=device, dtype=torch.float32).view(b - a, 1, 1, 1)
            block = base.unsqueeze(0) + drift.view(1, 1, 1, 3) * k + torch.randn((b - a, h, w, 3), device=device, generator=g) * 2.0
            if fade:
                block = block * torch.clamp(torch.minimum(k / 20.0, (length - 1 - k) / 20.0), 0.0, 1.0)
            x[t + a:t + b] = block.round().clamp(0, 255).to(torch.uint8)
        t += m
        shot += 1
    return x, cuts


class LazyClip:
    """Shape-only stand-in for a clip another rank owns (`distributed.score_clips_distributed` only touches its own)."""

    def __init__(self, n: int, h: int, w: int):
        self.shape = (n, h, w, 3)

    def __len__(self):
        return self.shape[0]


def make_packed_clips(specs, seeds, device, shot_len=(40, 400)):
    """Clips of `specs` [(n, h, w), ...]; clips of one resolution are carved out of ONE allocation, back to back, so the
    engine scores them in place with one launch per term (`ScoringEngine.score_clips`).  Returns (clips, truths) in the
    order of `specs`."""
    clips, truths = [None] * len(specs), [None] * len(specs)
    by_res: dict = {}
    for i, (n, h, w) in enumerate(specs):
        by_res.setdefault((h, w), []).append(i)
    for (h, w), idxs in by_res.items():
        total = sum(specs[i][0] for i in idxs)
        slab = torch.empty((total, h, w, 3), dtype=torch.uint8, device=device)
        off = 0
        for i in idxs:
            n = specs[i][0]
            clips[i], truths[i] = make_device_clip(seeds[i], n, h, w, device, shot_len, out=slab[off:off + n])
            off += n
    return clips, truths
