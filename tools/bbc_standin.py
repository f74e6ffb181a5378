"""BASELINE.json config 4 stand-in: AdaptiveDetector(window_width=2, min_content_val=15) over a set of
synthetic "broadcast" clips with known cut positions, sharded by clip over the GPUs of the node.

The BBC Planet Earth set (11 videos, reference benchmark/README.md:58-68) is not available offline and
there is no decoder in this image, so the clips are generated on the device: 640x360 shots of random
length built from a smooth random image plus noise and slow drift, hard cuts between shots, some shots
fading out and in through black.  Reports frames/s and a quick precision/recall/F1 against the generator's ground
truth; ``--dump FILE`` writes the predictions in the format of tools/bbc_harness.py (the reference's convention:
predicted cuts = the end of every scene, the last one included) for tools/bbc_evaluate.py, which scores them with the
reference's own benchmark/evaluator.py.

    python tools/bbc_standin.py [--clips 11] [--frames 6000] [--dump predictions.json]
    python -m torch.distributed.run --nproc-per-node 8 tools/bbc_standin.py   # sharded by clip
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.corpus import detect_corpus

H, W = 360, 640


def make_device_clip(seed, n, device):
    g = torch.Generator(device=device); g.manual_seed(seed)
    rs = np.random.default_rng(seed)
    x = torch.empty((n, H, W, 3), dtype=torch.uint8, device=device)
    cuts, t, shot = [], 0, 0
    while t < n:
        length = int(rs.integers(40, 400))
        grid = torch.rand((1, 3, 9, 16), device=device, generator=g) * 255.0
        base = torch.nn.functional.interpolate(grid, size=(H, W), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
        drift = torch.randn((3,), device=device, generator=g) * 0.05
        fade = shot % 5 == 4 and length > 60
        if shot:
            cuts.append(t)
        m = min(length, n - t)
        k = torch.arange(m, device=device, dtype=torch.float32).view(m, 1, 1, 1)
        block = base.unsqueeze(0) + drift.view(1, 1, 1, 3) * k + torch.randn((m, H, W, 3), device=device, generator=g) * 2.0
        if fade:
            gain = torch.clamp(torch.minimum(k / 20.0, (length - 1 - k) / 20.0), 0.0, 1.0)
            block = block * gain
        x[t:t + m] = block.round().clamp(0, 255).to(torch.uint8)
        t += m; shot += 1
    return x, cuts


def prf(pred, truth):
    tp = len(set(pred) & set(truth))
    p = tp / len(pred) if pred else 0.0
    r = tp / len(truth) if truth else 0.0
    return p, r, (2 * p * r / (p + r) if p + r else 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=11)
    ap.add_argument("--frames", type=int, default=6000)
    ap.add_argument("--dump", default=None, help="write predictions + ground truth for tools/bbc_evaluate.py")
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    eng = E.ScoringEngine(local)
    # every rank builds only the clips the greedy plan gives it (same plan as score_clips_distributed)
    from pyscenedetect_amd.distributed import assign_clips
    lengths = [args.frames + 137 * i for i in range(args.clips)]
    plan = assign_clips([n * H * W for n in lengths], world)

    class Lazy:  # shape-only stand-in for clips owned by other ranks
        def __init__(self, n): self.shape = (n, H, W, 3)
        def __len__(self): return self.shape[0]

    clips, truth = [], []
    for i, n in enumerate(lengths):
        if i in plan[rank]:
            x, cuts = make_device_clip(1000 + i, n, device)
        else:
            x, cuts = Lazy(n), None
        clips.append(x); truth.append(cuts)
    torch.cuda.synchronize()
    spec = {"adaptive": {"window_width": 2, "min_content_val": 15.0}}
    detect_corpus(eng, clips, 25.0, spec)           # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = detect_corpus(eng, clips, 25.0, spec)
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, {i: truth[i] for i in plan[rank]})
        for d in gathered:
            for i, c in d.items():
                truth[i] = c
    if rank == 0:
        pred = [r["adaptive"] for r in res]
        P = [prf(p, t) for p, t in zip(pred, truth)]
        allp = prf([(i, c) for i, p in enumerate(pred) for c in p], [(i, c) for i, t in enumerate(truth) for c in t])
        if args.dump:
            # scene ends as `detect()` reports them: every cut, then the end of the video (the annotations of the BBC
            # set list every scene, the last one too, benchmark/dataset.py:66-74)
            vids = [{"video_file": "standin_%02d" % i, "predicted_cuts": [int(c) for c in p] + ([lengths[i]] if p else []),
                     "hard_cuts": [int(c) for c in t] + [lengths[i]], "elapsed": dt * lengths[i] / sum(lengths)}
                    for i, (p, t) in enumerate(zip(pred, truth))]
            with open(args.dump, "w") as f:
                json.dump({"format": "psd-benchmark-predictions/1", "detector": "detect-adaptive", "n_gpus": world,
                           "dataset": "synthetic stand-in for BBC Planet Earth (%d clips of 640x360, %d frames)" % (args.clips, sum(lengths)),
                           "videos": vids}, f)
        print(json.dumps({"config": "AdaptiveDetector(w=2, min_content_val=15) on %d synthetic 640x360 clips, %d frames, sharded by clip"
                          % (args.clips, sum(lengths)), "n_gpus": world, "frames_per_s": round(sum(lengths) / dt, 1),
                          "seconds": round(dt, 4), "precision": round(allp[0], 4), "recall": round(allp[1], 4), "f1": round(allp[2], 4),
                          "true_cuts": sum(len(t) for t in truth), "detected": sum(len(p) for p in pred),
                          "per_clip_f1": [round(x[2], 3) for x in P]}))
    eng.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
