"""Every environment switch of libpsd_hip.so that reroutes work to another kernel, launch shape or host loop
("untested alternative code paths"): each setting runs tests/_switch_probe.py -- all entry points the
switch can touch, against the oracle -- in its own process, because the library reads the switches once per process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SWITCHES = [
    {},                                   # the defaults, through the same probe
    {"PSD_SCORE_DIRECT": "1"},            # register-staged loads instead of LDS-DMA on the fast path (psd_score_kernels.hip)
    {"PSD_SCORE_G": "2"},                 # two 16-pixel groups per lane in the small-workgroup HSV pass
    {"PSD_SCORE_G": "1"},                 # ... one group per lane everywhere (also the 16-wave fused pass)
    {"PSD_HSV_BLOCKS": "64"},             # few, long time walks
    {"PSD_HSV_BLOCKS": "1000000"},        # as many walks as frames allow
    {"PSD_LUMA_WALK": "1"},               # luma histogram on the time-walking kernel instead of luma_hist_kernel
    {"PSD_EDGE_FUSE_HSV": "0"},           # two reads of the frames for HSV + edges
    {"PSD_EDGE_WS_MB": "1"},              # edge workspace of a few frames: many chunks
    {"PSD_EDGE_WS_MB": "1", "PSD_EDGE_FUSE_HSV": "0"},
    {"PSD_SMALL_COPY": "1"},              # records of a small submission by one strided 2-D copy instead of a kernel's stores into the mirror
    {"PSD_SMALL_COPY": "2", "PSD_SPIN_US": "0"},   # ... packed + contiguous copy; completion by the runtime's wait instead of polling
    {"PSD_HASH_DIRECT": "1"},             # HashDetector thumbnails without LDS-DMA (psd_hash_kernels.hip)
    {"PSD_RESIZE_DEPTH": "3"},            # downscale kernel with two frames in flight
    {"PSD_RESIZE_ROWS": "1"},             # ... one destination row per workgroup
    {"PSD_RESIZE_ROWS": "5", "PSD_RESIZE_DEPTH": "3"},
    {"PSD_RESIZE_STORE_VEC": "0"},        # ... its storing instance with byte stores instead of 16-byte stores through LDS
    {"PSD_RESIZE_ROUNDS": "0"},           # ... time chunks by the rule of rounds 4-5 (12 workgroups per CU, rounded up)
    {"PSD_RESIZE_ROUNDS": "1"},           # ... one round: every workgroup resident at once, the longest walks
    {"PSD_EDGE_FUSE_DOWNSCALE": "0"},     # HSV + edges behind the downscale: resize into the buffer and read it again (the route until round 6)
    {"PSD_EDGE_FUSE_DOWNSCALE": "1", "PSD_EDGE_WS_MB": "1"},    # ... the fused front end over many small chunks
    {"PSD_CUBIC_FORM": "fma"},            # Interpolation.CUBIC as OpenCV builds with fused multiply-adds compute it (aarch64)
    {"PSD_CUBIC_FORM": "fixed"},          # ... as builds without the vector pass do (the scalar fixed point everywhere)
]


@pytest.mark.gpu
@pytest.mark.parametrize("switch", SWITCHES, ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()) or "defaults")
def test_switch_matches_the_oracle(switch):
    env = {k: v for k, v in os.environ.items() if not (k.startswith("PSD_") and k != "PSD_LIB_PATH")}
    env.update(switch)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_switch_probe.py")], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (switch, out.stdout[-400:], out.stderr[-1500:])
