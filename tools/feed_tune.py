"""SceneManager.detect_scenes (default pipeline: auto downscale, ContentDetector) over 1080p frames in pageable host memory:
frames/s by feeder settings.  One process per PSD_FEED_THREADS value (the library reads it once).
usage: PSD_FEED_THREADS=16 python tools/feed_tune.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import pyscenedetect_amd as psd  # noqa: E402
from pyscenedetect_amd import engine as E  # noqa: E402
from pyscenedetect_amd import scene_manager as smod  # noqa: E402

eng = E.ScoringEngine(0)
H, W, N = 1080, 1920, 512
rng = np.random.default_rng(0)
hf = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)


def run(stats: bool, batch_frames: int):
    sm = psd.SceneManager(psd.StatsManager() if stats else None, engine=eng, batch_frames=batch_frames)
    sm.add_detector(psd.ContentDetector(engine=eng))
    sm.detect_scenes(psd.ArrayVideoStream(hf, 25.0))


out = {"feed_threads": int(os.environ.get("PSD_FEED_THREADS", "16"))}
for feed_batch in (8, 16, 32):
    for slots in (3, 4):
        for bf in (64, 128):
            smod._DeviceFeeder.FEED_BATCH, smod._DeviceFeeder.N_SLOTS = feed_batch, slots
            for stats in (False, True):
                run(stats, bf)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    run(stats, bf)
                    best = min(best, time.perf_counter() - t0)
                out[f"feed{feed_batch}_slots{slots}_batch{bf}_{'stats' if stats else 'plain'}"] = round(N / best, 1)
print(json.dumps(out))
