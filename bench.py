"""bench.py -- frames/sec of the ContentDetector hot path on device-resident 1080p batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 4096] [--dist U|K] [--detector content|all]

Workload (BASELINE.json configs[1]): ContentDetector(threshold=27) over a synthetic
1920x1080 BGR batch of 4096 frames per GPU, resident in HBM when the timed region starts.
One "step" = one full pass of the hot path over the batch: the fused HIP scoring kernel,
the device->host copy of the per-frame records, and the native decision epilogue
(content_val + FlashFilter) that yields the cut list.  Steps are pipelined two deep
(the next kernel runs while the previous step's records are turned into cuts).

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank scores its own batch
(independent clips: weak scaling, no data-path collective) and the per-frame score vectors are
all-gathered over RCCL each step so every rank could run the epilogue for all clips.

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (device memory + torch.distributed only)

H, W = 1080, 1920
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is attainable
ALGO_BYTES_PER_PX = 3  # every BGR byte read once; edges would make it 5


def make_batch(n: int, dist: str, seed: int, device) -> torch.Tensor:
    x = torch.empty((n, H, W, 3), dtype=torch.uint8, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if dist == "U":  # i.i.d. uniform bytes: worst case for the hue branches, no DVFS give-back
        step = 64
        for i in range(0, n, step):
            x[i:i + step] = torch.randint(0, 256, x[i:i + step].shape, dtype=torch.uint8, device=device, generator=g)
    elif dist == "K":  # constant frames: worst case for histogram contention
        vals = torch.randint(0, 256, (n, 1, 1, 1), dtype=torch.uint8, device=device, generator=g)
        x.copy_(vals.expand_as(x))
    elif dist == "S":  # shots: smooth base image + per-frame noise, hard cuts every 64 frames
        shot = 64
        for s0 in range(0, n, shot):
            grid = torch.rand((1, 3, 9, 16), device=device, generator=g) * 255.0
            base = torch.nn.functional.interpolate(grid, size=(H, W), mode="bilinear", align_corners=True)
            base = base[0].permute(1, 2, 0)
            for i in range(s0, min(n, s0 + shot)):
                noise = torch.randn((H, W, 3), device=device, generator=g) * 2.0
                x[i] = (base + noise).round().clamp(0, 255).to(torch.uint8)
    else:
        raise ValueError(dist)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return x


def cpu_baseline(sample: np.ndarray, flags: int, threads: int) -> dict:
    """The CPU oracle (C restatement of the reference's cv2/numpy path) on a bounded sample of the
    same workload, all host cores (ctypes releases the GIL; disjoint frame ranges per thread,
    each with its one-frame halo)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import lib as orc

    n = len(sample)
    orc.score_batch(sample[:2], flags=flags)  # build/load + warm up

    def work(r):
        a, b = r
        return orc.score_batch(sample[a:b], sample[a - 1] if a > 0 else None, flags=flags)

    t0 = time.perf_counter()
    orc.score_batch(sample[: max(2, n // 8)], flags=flags)
    t1 = time.perf_counter() - t0
    single = max(2, n // 8) / t1
    bounds = [(i * n // threads, (i + 1) * n // threads) for i in range(threads)]
    bounds = [b for b in bounds if b[1] > b[0]]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, bounds))
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of the same batch through oracle/cv2_restate.c (gcc -O3), "
                      f"{threads} threads; single thread: {single:.2f} frames/s; real OpenCV is not installed",
            "_records": np.concatenate(parts)}


def main(argv=None, engine_factory=None, cpu_dry_run: bool = False) -> None:
    """``engine_factory``/``cpu_dry_run`` exist for tests/test_bench_plumbing.py only: they run this very
    control flow (pipelining, exchange, JSON) on CPU tensors over gloo with a stand-in engine, so the
    N > 1 path is exercised without GPUs.  The measured path always uses the HIP engine on cuda."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU batch")
    ap.add_argument("--dist", default="U", choices=["U", "K", "S"])
    ap.add_argument("--detector", default="content", choices=["content", "hist", "all", "hash"],
                    help="content = ContentDetector (headline); hist = Histogram+Threshold; all = all four fused; "
                         "hash = HashDetector (thumbnail kernel + DCT epilogue)")
    ap.add_argument("--res", default="1080p", choices=["1080p", "4k"])
    ap.add_argument("--cpu-sample", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--height", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--width", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    global H, W
    if args.res == "4k":
        H, W = 2160, 3840
    if args.height and args.width:
        H, W = args.height, args.width
    on_gpu = not cpu_dry_run

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Under torchrun (RANK set) the RCCL process group is always created, even for one rank, so the
    # exchange path below is the same code for N = 1 (launched that way) and N = 8.
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if on_gpu:
            torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when the communicator comes up; stdout must carry
        # exactly one JSON line, so point fd 1 at stderr until the first collective has run.
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if on_gpu:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                dist.barrier(device_ids=[local_rank])
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    device = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(device)

    from pyscenedetect_amd import engine as E
    from pyscenedetect_amd import epilogue

    eng = engine_factory(local_rank) if engine_factory else E.ScoringEngine(local_rank)
    flags = {"content": E.SCORE_HSV_SAD, "hist": E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM,
             "all": E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM, "hash": 0}[args.detector]
    n = args.frames
    batch = make_batch(n, args.dist, 20250921 + rank, device)
    ptr = batch.data_ptr()

    kernel_ms: list[float] = []
    state = {"cuts": None, "pending_gather": None, "gathered": None}

    def consume_gather():
        """Finish the score-vector all-gather issued one step earlier (keeps ranks loosely coupled)."""
        pend = state["pending_gather"]
        if pend is None:
            return
        work, recv, mine = pend
        work.wait()
        allv = recv.cpu().numpy().astype(np.uint64).reshape(world, -1, 4)   # every clip's score vectors
        assert np.array_equal(allv[rank, :, 0], mine)
        state["gathered"] = allv
        state["pending_gather"] = None

    def finish(collect_timing: bool):
        recs = eng.collect(n)
        if collect_timing:
            kernel_ms.append(eng.last_kernel_ms()[0])
        if use_dist:
            # score vectors only: 4 x u64 per frame
            import torch.distributed as dist

            consume_gather()
            vec = np.stack([recs["sad_h"], recs["sad_s"], recs["sad_v"], recs["edge_xor"]], axis=1)
            send = torch.from_numpy(vec.astype(np.int64)).to(device, non_blocking=True)
            recv = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=device)  # rank-major concat
            work = dist.all_gather_into_tensor(recv, send, async_op=True)
            state["pending_gather"] = (work, recv, recs["sad_h"].copy())
        state["recs"] = recs
        if args.detector in ("content", "all"):
            sc = epilogue.content_scores(recs, H, W)
            state["cuts"] = epilogue.content_cuts(sc["content_val"], 25.0, threshold=27.0, min_scene_len=15)
        if args.detector in ("hist", "all"):
            cuts, _ = epilogue.hist_cuts(recs, 25.0)
            epilogue.threshold_cuts(recs, H, W, 25.0)
            if args.detector == "hist":
                state["cuts"] = cuts
        if args.detector == "all":
            epilogue.adaptive_cuts(sc["content_val"], 25.0)

    def run_hash(steps: int, timing: bool):
        # HashDetector: grey INTER_AREA thumbnails on the device, DCT / median / Hamming distance / decision on the host
        for _ in range(steps):
            thumbs = eng.hash_thumbs_device(ptr, n, H, W, 16)
            if timing:
                kernel_ms.append(eng.last_kernel_ms()[0])
            bits = epilogue.hash_bits(thumbs, 8)
            state["cuts"] = epilogue.hash_cuts(bits, 25.0, threshold=0.35, min_scene_len=15)[0]
            state["thumbs"] = thumbs

    def run(steps: int, timing: bool):
        if args.detector == "hash":
            return run_hash(steps, timing)
        eng.submit_device(ptr, n, H, W, flags=flags)
        for _ in range(steps - 1):
            eng.submit_device(ptr, n, H, W, flags=flags)
            finish(timing)
        finish(timing)
        if use_dist:
            consume_gather()   # the last step's exchange completes inside the timed region

    def barrier():
        if use_dist:
            import torch.distributed as dist

            dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    if args.warmup > 0:
        run(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_frames = n * args.steps * world
    fps = total_frames / elapsed
    avg_kernel_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    algo_bytes = n * H * W * ALGO_BYTES_PER_PX
    achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                key = f"{args.detector}_{args.res}_{args.dist}_{n}"
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "frames/sec (1080p ContentDetector)" if (args.detector, args.res) == ("content", "1080p")
                      else f"frames/sec ({args.res} {args.detector})",
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": (f"ContentDetector(threshold=27) on {n} x 1920x1080 BGR frames per GPU, device-resident "
                             f"(BASELINE.json configs[1]); distribution {args.dist}")
                            if (args.detector, args.res) == ("content", "1080p") else
                            f"detector set '{args.detector}' on {n} x {W}x{H} BGR frames per GPU, device-resident; "
                            f"distribution {args.dist}",
                "frames_per_gpu": n, "height": H, "width": W, "distribution": args.dist,
                "parallelism": f"clips sharded over {world} GPU(s), RCCL all-gather of score vectors" if use_dist else "1 GPU",
                "pipeline_depth": 1 if args.detector == "hash" else 2,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": {"content": "psd::score_frames_dma_kernel", "all": "psd::score_frames_dma_kernel",
                           "hist": "psd::luma_hist_kernel", "hash": "psd::gray_area_dma_kernel"}[args.detector],
                "avg_launch_ms": round(avg_kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "cuts_found": len(state["cuts"]),
        }
        if not args.no_cpu_baseline and world == 1 and args.detector == "hash":
            from oracle import lib as orc

            sample = batch[: min(64, n)].cpu().numpy()
            t0 = time.perf_counter()
            ref = orc.hash_thumbs(sample, 16)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(len(sample) / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same batch through oracle/cv2_restate.c (grey + INTER_AREA), "
                                             "one thread; real OpenCV is not installed" % len(sample)}
            out["parity_sample"] = ("thumbnails of the first %d frames identical to the oracle" % len(sample)
                                    if np.array_equal(state["thumbs"][: len(sample)], ref) else "MISMATCH vs oracle")
        elif not args.no_cpu_baseline and world == 1:
            sample = batch[: args.cpu_sample].cpu().numpy()
            cb = cpu_baseline(sample, flags & 7, os.cpu_count() or 1)
            ref = cb.pop("_records")
            got = state["recs"][: args.cpu_sample]
            same = all(np.array_equal(got[f], ref[f]) for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"))
            out["cpu_baseline"] = cb
            out["parity_sample"] = "records of the first %d frames identical to the oracle" % args.cpu_sample if same \
                else "MISMATCH vs oracle"
        else:
            out["cpu_baseline"] = None
    eng.close()
    if use_dist:
        import torch.distributed as dist

        dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
