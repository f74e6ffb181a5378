"""Secondary measurements (not the headline): edge term, PCIe-inclusive host path, mixed corpus,
per-frame API latency.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pyscenedetect_amd as psd
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.corpus import detect_corpus

eng = E.ScoringEngine(0)
out = {}
H, W = 1080, 1920
# 1. edge term, device resident
N = 128
x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
for name, fl in (("hsv+edges", E.SCORE_HSV_SAD | E.SCORE_EDGES), ("all+edges", E.SCORE_ALL)):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); eng.score_device(x.data_ptr(), N, H, W, flags=fl); best = min(best, time.perf_counter() - t0)
    out[f"edges_1080p_uniform_{name}_fps"] = round(N / best, 1)
# smoother content (fewer edges) -- a shot-like batch
base = torch.nn.functional.interpolate(torch.rand((1, 3, 9, 16), device="cuda") * 255, size=(H, W), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
for i in range(N):
    x[i] = (base + torch.randn((H, W, 3), device="cuda") * 2).round().clamp(0, 255).to(torch.uint8)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); eng.score_device(x.data_ptr(), N, H, W, flags=E.SCORE_HSV_SAD | E.SCORE_EDGES); best = min(best, time.perf_counter() - t0)
out["edges_1080p_smooth_hsv+edges_fps"] = round(N / best, 1)
del x
# 2. host path (pageable numpy frames), PCIe inclusive
rng = np.random.default_rng(0)
hf = rng.integers(0, 256, (192, H, W, 3), dtype=np.uint8)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); eng.score_host(hf, flags=E.SCORE_HSV_SAD); best = min(best, time.perf_counter() - t0)
out["host_path_1080p_fps_pcie_inclusive"] = round(len(hf) / best, 1)
out["host_path_GBps"] = round(hf.nbytes / best / 1e9, 2)
# 3. per-frame API latency
det = psd.ContentDetector(engine=eng)
t0 = time.perf_counter()
for i in range(32):
    det.process_frame(psd.FrameTimecode(i, 25.0), hf[i])
out["process_frame_1080p_ms"] = round((time.perf_counter() - t0) / 32 * 1e3, 3)
# 3b. SceneManager end to end from host frames (decode thread -> batches -> device -> decisions)
for name, auto in (("full_res", False), ("auto_downscale", True)):
    sm = psd.SceneManager(engine=eng)
    sm.auto_downscale = auto
    sm.add_detector(psd.ContentDetector(engine=eng))
    t0 = time.perf_counter()
    nproc = sm.detect_scenes(psd.ArrayVideoStream(hf, 25.0))
    out[f"scene_manager_1080p_{name}_fps"] = round(nproc / (time.perf_counter() - t0), 1)
# 4. mixed corpus, all four detectors, one GPU
corpus = [hf[:96], hf[96:], rng.integers(0, 256, (24, 2160, 3840, 3), dtype=np.uint8), hf[:64]]
t0 = time.perf_counter()
res = detect_corpus(eng, corpus, 25.0, {"content": {}, "adaptive": {}, "hist": {}, "threshold": {}})
dt = time.perf_counter() - t0
out["mixed_corpus_all_four_frames"] = int(sum(len(c) for c in corpus))
out["mixed_corpus_all_four_fps_pcie_inclusive"] = round(sum(len(c) for c in corpus) / dt, 1)
print(json.dumps(out))
