"""Runs bench.main() with a CPU stand-in engine over gloo (spawned by test_bench_plumbing.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ctypes  # noqa: E402

import numpy as np  # noqa: E402

import bench  # noqa: E402
from oracle import lib as orc  # noqa: E402


class StandInEngine:
    """submit/collect face of ScoringEngine, scoring host memory with the CPU oracle."""

    def __init__(self, device):
        self._pending = []

    def submit_device(self, ptr, n, h, w, row_stride=None, frame_stride=None, d_prev=None, flags=7, edge_kernel=0, stream=None):
        buf = (ctypes.c_uint8 * (n * h * w * 3)).from_address(ptr)
        frames = np.frombuffer(buf, np.uint8).reshape(n, h, w, 3)
        self._pending.append(orc.score_batch(frames, flags=flags & 7))

    def score_device(self, ptr, n, h, w, row_stride=None, frame_stride=None, d_prev=None, flags=7, edge_kernel=0, stream=None):
        self.submit_device(ptr, n, h, w, flags=flags)
        return self.collect(n)

    def score_device_downscaled(self, ptr, n, h, w, dst_h, dst_w, frame_stride=None, d_prev=None, flags=1, edge_kernel=0, interpolation=1,
                                stream=None):
        shim = os.path.join(ROOT, "oracle", "cv2_shim")
        if shim not in sys.path:
            sys.path.append(shim)
        import cv2  # the oracle's shim

        buf = (ctypes.c_uint8 * (n * h * w * 3)).from_address(ptr)
        frames = np.frombuffer(buf, np.uint8).reshape(n, h, w, 3)
        small = np.stack([cv2.resize(f, (dst_w, dst_h), interpolation=interpolation) for f in frames])
        return orc.score_batch(small, flags=flags & 7)

    def collect(self, n, sums_only=False):
        rec = self._pending.pop(0)
        assert len(rec) == n
        if sums_only:
            from pyscenedetect_amd.engine import _sums_of

            return _sums_of(rec)
        return rec

    def last_kernel_ms(self):
        return 1.0, 1

    def close(self):
        pass


if __name__ == "__main__":
    bench.main(sys.argv[1:], engine_factory=StandInEngine, cpu_dry_run=True)
