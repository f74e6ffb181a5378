"""GPU: BASELINE.json configs[3] and [4] as FLOWS through the HIP engine, cut lists against the CPU oracle.

Round-2 review: the mixed 1080p + 4K corpus (all four detectors, clips packed per resolution) and the benchmark harness
of the reference (``benchmark/__main__.py:44-61``: one default detector per video through ``SceneManager``) had only
been run on CPU stand-ins.  Here they run through ``libpsd_hip.so`` at the real frame sizes and every cut list is
compared with the one the oracle engine yields for the same frames; plus thicker oracle samples on the configurations
BASELINE names (4K Histogram + Threshold for uniform and constant frames, shot-like 1080p across a time-walk chunk
boundary).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import lib as orc
from oracle.detectors_np import OracleEngine
from pyscenedetect_amd import corpus, epilogue
from pyscenedetect_amd import engine as E

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
ALL_FOUR = {"content": {"min_scene_len": 6}, "adaptive": {"min_scene_len": 6}, "hist": {"min_scene_len": 6},
            "threshold": {"threshold": 40, "min_scene_len": 6}}


def oracle_records(frames: np.ndarray, flags: int = 7, threads: int = 16) -> np.ndarray:
    """orc.score_batch over disjoint frame ranges (one-frame halo each), a few threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    n = len(frames)
    bounds = [(i * n // threads, (i + 1) * n // threads) for i in range(threads)]
    bounds = [b for b in bounds if b[1] > b[0]]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda r: orc.score_batch(frames[r[0]:r[1]], frames[r[0] - 1] if r[0] else None, flags=flags), bounds))
    return np.concatenate(parts)


def test_detect_corpus_mixed_1080p_4k_all_four_detectors(hip_engine):
    """configs[4] in miniature but at the real frame sizes: 1080p and 4K clips, device-resident (packed back to back per
    resolution) and host clips mixed, all four detectors from the fused pass == the oracle engine's cut lists."""
    import torch

    import device_clips as DC

    dev = torch.device("cuda", 0)
    specs = [(40, 1080, 1920), (18, 2160, 3840), (26, 1080, 1920), (33, 1080, 1920), (20, 2160, 3840)]
    clips, truth = DC.make_packed_clips(specs, [70 + i for i in range(len(specs))], dev, shot_len=(7, 14))
    torch.cuda.synchronize()
    host = [c.cpu().numpy() for c in clips]
    mixed = [clips[0], clips[1], host[2], clips[3], host[4]]          # device and host clips side by side
    got = corpus.detect_corpus(hip_engine, mixed, 25.0, ALL_FOUR)
    want = corpus.detect_corpus(OracleEngine(), host, 25.0, ALL_FOUR)
    assert got == want
    assert all(r["content"] for r in want), "the clips have hard cuts: every ContentDetector list must be non-empty"
    for r, t in zip(want, truth):
        assert set(r["content"]) <= set(t) | {c + 1 for c in t}, "ContentDetector cuts sit on the generator's shot starts"
    # the records behind the decisions, clip by clip (packed scoring == per-clip scoring == oracle)
    flags = corpus.required_flags(ALL_FOUR)
    recs = corpus.score_clips(hip_engine, mixed, flags)
    for r, h in zip(recs, host):
        w = oracle_records(h, flags)
        for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"):
            assert np.array_equal(r[f], w[f]), f


def test_detect_corpus_with_the_edge_term(hip_engine):
    """ContentDetector weights (1,1,1,1) inside the corpus flow at 1080p (k = 13) and 720p: edge_xor through packed clips."""
    import torch

    import device_clips as DC

    spec = {"content": {"weights": (1.0, 1.0, 1.0, 1.0), "min_scene_len": 4, "threshold": 20.0}}
    specs = [(9, 1080, 1920), (7, 720, 1280), (6, 1080, 1920)]
    clips, _ = DC.make_packed_clips(specs, [90, 91, 92], torch.device("cuda", 0), shot_len=(3, 5))
    for c in clips:                                                  # something with edges that moves
        for t in range(len(c)):
            c[t, 100 + 9 * t:400 + 9 * t, 200 + 14 * t:700 + 14 * t] = torch.tensor([230, 40, 200], dtype=torch.uint8, device=c.device)
    torch.cuda.synchronize()
    host = [c.cpu().numpy() for c in clips]
    assert corpus.detect_corpus(hip_engine, clips, 25.0, spec) == corpus.detect_corpus(OracleEngine(), host, 25.0, spec)
    flags = corpus.required_flags(spec)
    assert flags & E.SCORE_EDGES
    got = corpus.score_clips(hip_engine, clips, flags)
    from oracle.detectors_np import score_batch as oracle_score

    for r, h in zip(got, host):
        w = oracle_score(h, edges=True)
        assert np.array_equal(r["edge_xor"], w["edge_xor"]) and w["edge_xor"][1:].any()


def test_bbc_standin_flow_adaptive(hip_engine):
    """configs[3]: AdaptiveDetector(window_width=2, min_content_val=15) over device-generated 640x360 clips, sharded by
    clip (one rank here) == oracle cut lists == the generator's ground truth."""
    import torch

    import device_clips as DC

    spec = {"adaptive": {"window_width": 2, "min_content_val": 15.0}}
    specs = [(300 + 37 * i, 360, 640) for i in range(4)]
    clips, truth = DC.make_packed_clips(specs, [1000 + i for i in range(4)], torch.device("cuda", 0), shot_len=(20, 90))
    torch.cuda.synchronize()
    got = corpus.detect_corpus(hip_engine, clips, 25.0, spec)
    want = corpus.detect_corpus(OracleEngine(), [c.cpu().numpy() for c in clips], 25.0, spec)
    assert got == want
    hit = sum(len(set(r["adaptive"]) & set(t)) for r, t in zip(got, truth))
    assert hit >= 0.9 * sum(len(r["adaptive"]) for r in got) > 0


def test_benchmark_harness_through_hip(tmp_path, hip_engine, oracle_engine):
    """tools/bbc_harness.run_predictions (the reference's `_run_predictions`, benchmark/__main__.py:44-61) on .npy clips in
    the BBC layout (benchmark/dataset.py:77-106): default-constructed detectors through SceneManager with its default
    downscale, HIP engine == oracle engine, for every detector of the reference's table."""
    import bbc_harness as H
    from pyscenedetect_amd.synth import make_clip

    os.makedirs(tmp_path / "videos")
    os.makedirs(tmp_path / "fixed")
    truths = {}
    for vid, (seed, n) in {"01": (7, 150), "02": (8, 131)}.items():
        frames, cuts = make_clip(seed, n, 360, 640, shot_len=(17, 40))
        np.save(tmp_path / "videos" / f"bbc_{vid}.npy", frames)
        bounds = [0, *cuts, n]
        truths[vid] = [*cuts, n]
        with open(tmp_path / "fixed" / f"{vid}-scenes.txt", "w") as f:
            for a, b in zip(bounds[:-1], bounds[1:]):
                f.write(f"{a}\t{b - 1}\n")
    samples = H.bbc_samples(str(tmp_path))
    assert [s["hard_cuts"] for s in samples] == [truths["01"], truths["02"]]
    for det in ("detect-adaptive", "detect-content", "detect-hist", "detect-threshold", "detect-hash"):
        got = H.run_predictions(samples, det, engine=hip_engine)
        want = H.run_predictions(samples, det, engine=oracle_engine)
        assert [r["predicted_cuts"] for r in got] == [r["predicted_cuts"] for r in want], det
        if det in H.PACKED:      # the whole dataset in shared device batches (detect_corpus behind the default downscale): the same predictions
            packed = H.run_predictions_packed(samples, det, hip_engine)
            assert [r["predicted_cuts"] for r in packed] == [r["predicted_cuts"] for r in want], det
        if det in ("detect-adaptive", "detect-content"):
            assert all(len(r["predicted_cuts"]) > 1 for r in got), det


def test_detect_corpus_under_a_one_rank_rccl_group():
    """The sharded flow on the GPU with a real RCCL process group (one rank: the only size a 1-GPU box offers): plan,
    packed scoring, all-gather of the records, decisions == the same call without a process group.  Round 6: the exchange is the
    one the C-ABI advertises (``psd_comm_*`` / ``psd_allgather_host``: RCCL loaded by ``libpsd_hip.so``, counts from the plan, one
    collective, no torch staging); ``PSD_NATIVE_EXCHANGE=0`` keeps ``torch.distributed``'s."""
    code = r'''
import os, sys, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() %% 200), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
import device_clips as DC
from pyscenedetect_amd import corpus, engine as E, distributed as D
eng = E.ScoringEngine(0)
spec = {"content": {"min_scene_len": 6}, "adaptive": {"min_scene_len": 6}, "hist": {}, "threshold": {}}
clips, _ = DC.make_packed_clips([(60, 360, 640), (25, 1080, 1920), (45, 360, 640)], [3, 4, 5], torch.device("cuda", 0), shot_len=(8, 20))
plain = corpus.detect_corpus(eng, clips, 25.0, spec)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
flags = corpus.required_flags(spec)
recs = D.score_clips_distributed(eng, clips, flags)
ok = all((a[f] == b[f]).all() for a, b in zip(recs, corpus.score_clips(eng, clips, flags)) for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"))
comm = D.native_comm_for(eng, None)                      # the exchange went through psd_allgather_host (RCCL loaded by libpsd_hip.so)
native = comm is not None and comm.exchanges == 1
def same(a, b, fields):
    return all((x[f] == y[f]).all() for x, y in zip(a, b) for f in fields)
sums = D.score_clips_distributed(eng, clips, E.SCORE_HSV_SAD, downscale="auto")      # 40-byte sums behind the default downscale
native = native and comm.exchanges == 2 and sums[0].dtype.itemsize == 40
ok_sums = same(sums, corpus.score_clips(eng, clips, E.SCORE_HSV_SAD, downscale="auto"), ("sad_h", "sad_s", "sad_v"))
os.environ["PSD_NATIVE_EXCHANGE"] = "0"                  # the torch.distributed path stays selectable
D._native_comms.clear()
recs_torch = D.score_clips_distributed(eng, clips, flags)
assert D.native_comm_for(eng, None) is None
same_cuts = same(recs_torch, recs, ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"))
grouped = corpus.detect_corpus(eng, clips, 25.0, spec)   # (one rank: detect_corpus keeps to its single-process path)
dist.destroy_process_group()
print("RESULT " + json.dumps({"same_records": bool(ok), "native": bool(native), "same_cuts": bool(same_cuts) and grouped == plain,
                              "sums": bool(ok_sums), "cuts": sum(len(v) for r in plain for v in r.values())}))
''' % (ROOT, ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    import json

    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert res["same_records"] and res["cuts"] > 0
    assert res["native"] and res["same_cuts"] and res["sums"]


# ---- thicker oracle samples on BASELINE's own configurations ----------------------------------------------------------

@pytest.mark.parametrize("dist_name", ["U", "K"])
def test_4k_histogram_threshold_16_frames_vs_oracle(hip_engine, oracle_engine, dist_name):
    """configs[2]: 16 frames of 3840x2160 through `luma_hist_kernel` ALONE (flags LUMA_HIST | BYTE_SUM) vs the oracle, for
    uniform bytes and for constant frames (one histogram bin per frame); then the cut lists of HistogramDetector with 128
    and 256 bins and ThresholdDetector(12), batch epilogues over HIP records == the frame-by-frame host detectors over the
    oracle engine (histogram_detector.py:98-165, threshold_detector.py:100-168)."""
    import pyscenedetect_amd as psd

    n, h, w = 16, 2160, 3840
    rng = np.random.default_rng(77)
    if dist_name == "U":
        frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        frames[5:8] //= 32                      # three dark frames: a fade for ThresholdDetector, a jump for the histograms
    else:
        vals = [200, 200, 3, 3, 3, 90, 90, 255, 0, 0, 17, 17, 17, 17, 128, 128]
        frames = np.empty((n, h, w, 3), np.uint8)
        for i, v in enumerate(vals):
            frames[i] = v
    buf = hip_engine.alloc(frames.nbytes)
    buf.upload(frames.reshape(-1))
    got = hip_engine.score_device(buf.ptr, n, h, w, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM)
    buf.free()
    want = oracle_records(frames, 6)
    assert np.array_equal(got["hist"], want["hist"]) and np.array_equal(got["byte_sum"], want["byte_sum"])
    assert (got["hist"].sum(axis=1) == h * w).all()
    for bins in (128, 256):
        sm = psd.SceneManager(engine=oracle_engine)
        sm.auto_downscale = False
        sm.add_detector(psd.HistogramDetector(bins=bins, min_scene_len=1, engine=oracle_engine))
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
        ref_cuts = [c.frame_num for c in sm.get_cut_list(show_warning=False)]
        cuts, _ = epilogue.hist_cuts(got, 25.0, 0.20, bins, 1)
        assert cuts == ref_cuts and len(cuts) >= 2, (bins, cuts, ref_cuts)
    sm = psd.SceneManager(engine=oracle_engine)
    sm.auto_downscale = False
    sm.add_detector(psd.ThresholdDetector(threshold=12, min_scene_len=1, engine=oracle_engine))
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    ref_cuts = [c.frame_num for c in sm.get_cut_list(show_warning=False)]
    cuts, _ = epilogue.threshold_cuts(got, h, w, 25.0, 12, 1)
    assert cuts == ref_cuts and cuts, (cuts, ref_cuts)


def test_shot_like_1080p_across_a_time_walk_chunk_boundary(hip_engine):
    """configs[1] on realistic content: 160 shot-like 1080p frames (the HSV pass walks chunks of about 63 frames, so the
    batch holds two chunk boundaries) -- the records of 72 frames around both boundaries against the oracle, all three
    non-edge terms, and the cut list of the whole batch."""
    import torch

    import device_clips as DC

    n, h, w = 160, 1080, 1920
    x, truth = DC.make_device_clip(99, n, h, w, torch.device("cuda", 0), shot_len=(9, 30))
    torch.cuda.synchronize()
    flags = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
    fused = hip_engine.score_device(x.data_ptr(), n, h, w, flags=flags)
    hsv = hip_engine.score_device(x.data_ptr(), n, h, w, flags=E.SCORE_HSV_SAD)
    for a, b in ((40, 88), (112, 136)):
        host = x[a - 1:b].cpu().numpy()
        want = orc.score_batch(host[1:], host[0])
        for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"):
            assert np.array_equal(fused[f][a:b], want[f]), (f, a, b)
        for f in ("sad_h", "sad_s", "sad_v"):
            assert np.array_equal(hsv[f][a:b], want[f]), (f, a, b)
    sc = epilogue.content_scores(hsv, h, w)
    cuts = epilogue.content_cuts(sc["content_val"], 25.0, min_scene_len=5)
    assert cuts and set(cuts) <= set(truth)
