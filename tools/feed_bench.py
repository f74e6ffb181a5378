"""End-to-end rates of the host-fed paths (PCIe inclusive; never the headline `value`): SceneManager.detect_scenes from
host frames (pageable and page-locked), psd_score_batch, and the pinned async upload on its own.  One JSON object."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import pyscenedetect_amd as psd  # noqa: E402
from pyscenedetect_amd import engine as E  # noqa: E402

eng = E.ScoringEngine(0)
out = {}
H, W, N = 1080, 1920, 384
rng = np.random.default_rng(0)
hf = rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)
pinned = eng.pinned_array((N, H, W, 3))
pinned[:] = hf


def best_of(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


out["psd_score_batch_pageable_fps"] = round(N / best_of(lambda: eng.score_host(hf, flags=E.SCORE_HSV_SAD)), 1)
out["psd_score_batch_pinned_fps"] = round(N / best_of(lambda: eng.score_host(pinned, flags=E.SCORE_HSV_SAD)), 1)
# the same entry point behind the default downscale (1080p -> 256 x 144): the stacked array travels as tap rows only
out["score_host_downscaled_pageable_fps"] = round(N / best_of(lambda: eng.score_host(hf, flags=E.SCORE_HSV_SAD, downscale=7.5)), 1)
out["score_host_downscaled_pinned_fps"] = round(N / best_of(lambda: eng.score_host(pinned, flags=E.SCORE_HSV_SAD, downscale=7.5)), 1)
# round 6: the packed flow on clips in HOST memory (corpus.detect_corpus: three clips of 128 frames, AdaptiveDetector), behind the reference's
# default downscale -- tap rows only -- and at full resolution (whole frames)
from pyscenedetect_amd import corpus  # noqa: E402

host_clips = [hf[0:128], hf[128:256], hf[256:384]]
for name, auto in (("auto_downscale", True), ("full_res", False)):
    out[f"detect_corpus_host_clips_{name}_pageable_fps"] = round(
        N / best_of(lambda: corpus.detect_corpus(eng, host_clips, 25.0, {"adaptive": {}}, auto_downscale=auto)), 1)
buf = eng.alloc(hf.nbytes)


def up_async():
    eng.upload_async(buf.ptr, pinned)
    eng.upload_fence(wait_on_host=True)


dt = best_of(up_async)
out["upload_async_pinned_GBps"] = round(hf.nbytes / dt / 1e9, 2)
dt = best_of(lambda: buf.upload_unordered(hf.reshape(-1)))
out["upload_blocking_pageable_GBps"] = round(hf.nbytes / dt / 1e9, 2)
# the default pipeline's upload: 288 of 1080 rows (1080p -> 256 x 144), two strided copies per frame (psd_upload_rows)
rows = eng.downscale_source_rows(H, W, 144, 256, 1)
out["tap_rows_per_frame"] = int(len(rows))
out["tap_row_copies_per_frame"] = int(len(eng.upload_rows_plan(rows)))
for src_name, src in (("pageable", hf), ("pinned", pinned)):
    def up_rows():
        for t in range(N):
            buf.upload_rows(src[t], t * H * W * 3, rows)
    out[f"upload_tap_rows_{src_name}_fps"] = round(N / best_of(up_rows), 1)

    def up_whole():
        for t in range(N):
            buf.upload_unordered(src[t].reshape(-1), t * H * W * 3)
    out[f"upload_whole_frames_{src_name}_fps"] = round(N / best_of(up_whole), 1)
# ABI 5: the same rows, 16 frames per call, gathered by the engine's worker threads into page-locked memory + one async copy
out["feed_threads"] = int(os.environ.get("PSD_FEED_THREADS", "16"))
for per_call in (8, 16, 32):
    views = [hf[t] for t in range(N)]

    def up_batched():
        for a in range(0, N, per_call):
            buf.upload_rows_batch(views[a:a + per_call], a * H * W * 3, rows, H * W * 3)
        eng.upload_fence(wait_on_host=True)
    out[f"upload_tap_rows_batched_{per_call}_pageable_fps"] = round(N / best_of(up_batched), 1)
for src_name, src in (("pageable", hf), ("pinned", pinned)):
    for name, auto in (("full_res", False), ("auto_downscale", True)):
        def run():
            sm = psd.SceneManager(engine=eng)
            sm.auto_downscale = auto
            sm.add_detector(psd.ContentDetector(engine=eng))
            sm.detect_scenes(psd.ArrayVideoStream(src, 25.0))
        out[f"scene_manager_1080p_{name}_{src_name}_fps"] = round(N / best_of(run, 3), 1)
print(json.dumps(out))
