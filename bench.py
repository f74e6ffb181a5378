"""bench.py -- frames/sec of the ContentDetector hot path on device-resident 1080p batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 4096] [--dist U|S|K]
                    [--detector content|hist|all|hash] [--res 1080p|4k] [--downscale auto|F] [--no-secondary]

Workload (BASELINE.json configs[1]): ContentDetector(threshold=27) over a synthetic
1920x1080 BGR batch of 4096 frames per GPU, resident in HBM when the timed region starts.
One "step" = one full pass of the hot path over the batch: the HIP scoring kernel,
the device->host copy of the per-frame records, and the native decision epilogue
(content_val + FlashFilter) that yields the cut list.  Steps are pipelined two deep
(the next kernel runs while the previous step's records are turned into cuts).

--downscale auto puts the reference's default downscale in front (SceneManager.auto_downscale,
scene_manager.py:110,123-140,666-678: cv2.resize to about 256 pixels width, INTER_LINEAR); the roofline then counts the
source rows that carry taps (2 per destination row), which is all that pipeline has to read.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank scores its own batch
(independent clips: weak scaling, no data-path collective) and the per-frame score vectors are
all-gathered over RCCL each step straight from the device-resident records, so every rank could run the
epilogue for all clips.

Prints ONE JSON line on rank 0.  At N = 1 the line also carries `secondary`: short runs of the other BASELINE
configurations (4K Histogram + Threshold, all four detectors fused, shot-like and constant content, the default
downscaled pipeline, the edge term, HashDetector) so that one driver run records them all.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is attainable
ALGO_BYTES_PER_PX = 3  # every BGR byte read once; edges make it 5


def make_batch(n: int, dist: str, seed: int, device, h: int, w: int) -> torch.Tensor:
    x = torch.empty((n, h, w, 3), dtype=torch.uint8, device=device)
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
            base = torch.nn.functional.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)
            base = base[0].permute(1, 2, 0)
            for i in range(s0, min(n, s0 + shot)):
                noise = torch.randn((h, w, 3), device=device, generator=g) * 2.0
                x[i] = (base + noise).round().clamp(0, 255).to(torch.uint8)
    else:
        raise ValueError(dist)
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return x


def downscaled_size(h: int, w: int, downscale: str) -> tuple[int, int, float]:
    """Target size exactly as the reference computes it (scene_manager.py:123-140, 670-678)."""
    factor = max(h, w) / 256.0 if downscale == "auto" else float(downscale)
    if downscale == "auto" and max(h, w) < 256:
        factor = 1.0
    return max(1, round(h / factor)), max(1, round(w / factor)), factor


def cpu_baseline(sample: np.ndarray, flags: int, threads: int) -> dict:
    """The CPU side of the same workload on this box's host cores, bounded samples:
      * value / cores: the C oracle (restatement of the reference's cv2 / numpy pixel path) over all host cores
        (ctypes releases the GIL; disjoint frame ranges per thread, each with its one-frame halo);
      * reference_model: the reference's own execution model -- one Python process, frame by frame, numpy for what it
        does in numpy and the cv2 restatement for what it does in OpenCV (oracle/reference_loop.py), plus the
        numpy-only half on its own (that half IS the reference's code: content_detector.py:29-36)."""
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
    out = {"value": round(n / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{n} frames of the same batch through oracle/cv2_restate.c (gcc -O3), "
                     f"{threads} threads; single thread: {single:.2f} frames/s; real OpenCV is not installed",
           "single_thread_frames_per_s": round(single, 2),
           "_records": np.concatenate(parts)}
    if flags & 1:
        shim = os.path.join(ROOT, "oracle", "cv2_shim")
        if shim not in sys.path:
            sys.path.append(shim)
        from oracle.reference_loop import time_models

        m = time_models(sample[: min(n, 48)])
        m.pop("_scores")
        out["reference_model"] = {
            "what": "ContentDetector.process_frame as the reference runs it (scene_manager.py:578-585): one Python process, "
                    "one frame at a time, cv2.cvtColor + cv2.split (C restatement) then 3x _mean_pixel_distance (the "
                    "reference's numpy expression, content_detector.py:29-36)",
            "frames_per_s": m["python_loop_frames_per_s"], "cores": 1, "frames": m["python_loop_frames"],
            "numpy_half_only_frames_per_s": m["numpy_half_frames_per_s"],
        }
    return out


class Workload:
    """One configuration of the hot path on a resident batch: how to submit a step, how to finish it, what it reads."""

    def __init__(self, eng, batch, detector: str, downscale: str | None, epilogue_mod, E, stream=None):
        self.eng, self.batch, self.detector, self.downscale, self.ep, self.E = eng, batch, detector, downscale, epilogue_mod, E
        self.n, self.h, self.w = batch.shape[0], batch.shape[1], batch.shape[2]
        self.ptr = batch.data_ptr()
        self.stream = stream
        self.flags = {"content": E.SCORE_HSV_SAD, "hist": E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM,
                      "all": E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM, "hash": 0,
                      "edges": E.SCORE_HSV_SAD | E.SCORE_EDGES}[detector]
        self.sh, self.sw = self.h, self.w                  # size the detectors see
        if downscale:
            self.sh, self.sw, self.factor = downscaled_size(self.h, self.w, downscale)
        self.state = {"cuts": [], "recs": None, "thumbs": None}

    # bytes one launch has to read: every BGR byte once; with the edge term 5 B/px; behind the default
    # downscale the source rows that carry taps (two per destination row, never more than all rows)
    def algorithmic_bytes(self) -> int:
        if self.downscale:
            rows = min(self.h, 2 * self.sh)
            return self.n * rows * self.w * 3
        per_px = 5 if self.detector == "edges" else ALGO_BYTES_PER_PX
        return self.n * self.h * self.w * per_px

    def kernel_name(self) -> str:
        if self.downscale:
            return "psd::resize_walk_kernel"
        return {"content": "psd::score_frames_dma_kernel", "all": "psd::score_frames_dma_kernel", "hist": "psd::luma_hist_kernel",
                "hash": "psd::gray_area_dma_kernel", "edges": "psd::sobel_nms_tile_kernel (edge pipeline) + psd::score_frames_dma_kernel"}[self.detector]

    def submit(self):
        if self.downscale:
            self.eng.submit_device_downscaled(self.ptr, self.n, self.h, self.w, self.sh, self.sw, flags=self.flags)
        elif self.stream is not None:
            self.eng.submit_device(self.ptr, self.n, self.h, self.w, flags=self.flags, stream=self.stream)
        else:
            self.eng.submit_device(self.ptr, self.n, self.h, self.w, flags=self.flags)

    def finish(self) -> float:
        """Collect one submission and turn its records into cut lists; returns the kernel time (ms)."""
        recs = self.eng.collect(self.n)
        ms = self.eng.last_kernel_ms()[0]
        ep, st = self.ep, self.state
        st["recs"] = recs
        if self.detector in ("content", "all", "edges"):
            weights = (1.0, 1.0, 1.0, 1.0) if self.detector == "edges" else (1.0, 1.0, 1.0, 0.0)
            sc = ep.content_scores(recs, self.sh, self.sw, weights)
            st["cuts"] = ep.content_cuts(sc["content_val"], 25.0, threshold=27.0, min_scene_len=15)
        if self.detector in ("hist", "all"):
            cuts, _ = ep.hist_cuts(recs, 25.0)
            ep.threshold_cuts(recs, self.sh, self.sw, 25.0)
            if self.detector == "hist":
                st["cuts"] = cuts
        if self.detector == "all":
            ep.adaptive_cuts(sc["content_val"], 25.0)
        return ms

    def run_hash_step(self) -> float:
        # HashDetector: grey INTER_AREA thumbnails on the device, DCT / median / Hamming distance / decision on the host
        thumbs = self.eng.hash_thumbs_device(self.ptr, self.n, self.h, self.w, 16)
        ms = self.eng.last_kernel_ms()[0]
        bits = self.ep.hash_bits(thumbs, 8)
        self.state["cuts"] = self.ep.hash_cuts(bits, 25.0, threshold=0.35, min_scene_len=15)[0]
        self.state["thumbs"] = thumbs
        return ms


def quick_measure(wl: Workload, steps: int = 5, warmup: int = 2) -> dict:
    """Short single-GPU measurement of a secondary workload: same step definition, two steps in flight."""
    def run(k, sink):
        if wl.detector == "hash":
            for _ in range(k):
                sink.append(wl.run_hash_step())
            return
        wl.submit()
        for _ in range(k - 1):
            wl.submit()
            sink.append(wl.finish())
        sink.append(wl.finish())

    run(warmup, [])
    torch.cuda.synchronize()
    ms: list[float] = []
    t0 = time.perf_counter()
    run(steps, ms)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(ms))
    achieved = wl.algorithmic_bytes() / (k_ms * 1e-3) / 1e9
    return {"value": round(wl.n * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "avg_launch_ms": round(k_ms, 4), "achieved_GBps": round(achieved, 1), "frac_of_8TBps": round(achieved / HBM_PEAK_GBS, 4),
            "kernel": wl.kernel_name(), "cuts_found": len(wl.state["cuts"])}


def secondary_runs(eng, batch, E, epilogue, device, frames_small: int) -> dict:
    """The other BASELINE configurations, a few steps each (N = 1 only).  Failures are recorded, never fatal."""
    out = {}
    h, w = batch.shape[1], batch.shape[2]

    def attempt(name, workload, fn):
        try:
            r = fn()
            r["workload"] = workload
            out[name] = r
        except Exception as ex:  # noqa: BLE001 -- the headline line must survive a broken secondary
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:300], "workload": workload}

    n = batch.shape[0]
    attempt("all_four_detectors_fused_1080p", f"Content + Adaptive + Histogram + Threshold from one pass, {n} x {w}x{h}, uniform bytes",
            lambda: quick_measure(Workload(eng, batch, "all", None, epilogue, E)))
    attempt("default_pipeline_downscale_auto", f"ContentDetector behind SceneManager's default downscale ({w}x{h} -> 256 wide), {n} frames; "
            "roofline counts the source rows that carry taps", lambda: quick_measure(Workload(eng, batch, "content", "auto", epilogue, E)))
    attempt("hash_detector_1080p", f"HashDetector (thumbnail kernel + DCT epilogue), {n} x {w}x{h}",
            lambda: quick_measure(Workload(eng, batch, "hash", None, epilogue, E), steps=3, warmup=1))
    for dist, label in (("S", "shot-like content (64-frame shots, hard cuts)"), ("K", "constant frames (one histogram bin per frame)")):
        def run_dist(dist=dist):
            b = make_batch(frames_small, dist, 20250921, device, h, w)
            r = quick_measure(Workload(eng, b, "content", None, epilogue, E))
            if dist == "S":
                r["edges_weights_1111"] = quick_measure(Workload(eng, b, "edges", None, epilogue, E), steps=3, warmup=1)
                r["all_four_fused"] = quick_measure(Workload(eng, b, "all", None, epilogue, E))
            else:
                r["histogram_threshold"] = quick_measure(Workload(eng, b, "hist", None, epilogue, E))
            del b
            return r
        attempt(f"content_1080p_{dist}", f"ContentDetector, {frames_small} x {w}x{h}, {label}", run_dist)

    def run_4k():
        b = make_batch(frames_small, "U", 20250921, device, 2160, 3840)
        r = quick_measure(Workload(eng, b, "hist", None, epilogue, E))
        r["content_detector"] = quick_measure(Workload(eng, b, "content", None, epilogue, E))
        del b
        return r
    attempt("histogram_threshold_4k", f"BASELINE configs[2]: HistogramDetector + ThresholdDetector, {frames_small} x 3840x2160, uniform bytes", run_4k)
    torch.cuda.empty_cache()
    return out


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
    ap.add_argument("--detector", default="content", choices=["content", "hist", "all", "hash", "edges"],
                    help="content = ContentDetector (headline); hist = Histogram+Threshold; all = all four fused; "
                         "hash = HashDetector (thumbnail kernel + DCT epilogue); edges = ContentDetector with weights (1,1,1,1)")
    ap.add_argument("--res", default="1080p", choices=["1080p", "4k"])
    ap.add_argument("--downscale", default=None, help="'auto' (the reference's default: to about 256 px width) or a factor")
    ap.add_argument("--cpu-sample", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--secondary-frames", type=int, default=2048)
    ap.add_argument("--height", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--width", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    H, W = (2160, 3840) if args.res == "4k" else (1080, 1920)
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
    n = args.frames
    batch = make_batch(n, args.dist, 20250921 + rank, device, H, W)
    wl = Workload(eng, batch, args.detector, args.downscale, epilogue, E)

    kernel_ms: list[float] = []
    state = {"pending_gather": None, "gathered": None}

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

    def score_vectors_on_device(recs):
        """int64[n, 4] (sad_h, sad_s, sad_v, edge_xor) on the device: a strided view of the records where the kernels
        left them (psd_last_records_device) -- no host bounce; the stand-in engine of the CPU dry run has no device."""
        if on_gpu and hasattr(eng, "last_records_device") and not state.get("no_device_view"):
            try:
                ptr, cnt = eng.last_records_device()
                assert cnt == n

                class _Recs:  # zero-copy view of the engine's record slot: n x 133 int64 (1064 bytes per record)
                    __cuda_array_interface__ = {"shape": (n, 133), "typestr": "<i8", "data": (ptr, False), "version": 2}

                return torch.as_tensor(_Recs(), device=device)[:, :4].contiguous()
            except Exception as ex:  # noqa: BLE001 -- never lose a scaling run to the view: fall back to the host copy
                state["no_device_view"] = f"{type(ex).__name__}: {ex}"
                print("bench: device view of the records unavailable (%s); exchanging the host copy" % state["no_device_view"], file=sys.stderr)
        vec = np.stack([recs["sad_h"], recs["sad_s"], recs["sad_v"], recs["edge_xor"]], axis=1)
        return torch.from_numpy(vec.astype(np.int64)).to(device)

    def finish(collect_timing: bool):
        ms = wl.finish()
        if collect_timing:
            kernel_ms.append(ms)
        if use_dist:
            # score vectors only: 4 x u64 per frame, all-gathered from HBM
            import torch.distributed as dist

            consume_gather()
            recs = wl.state["recs"]
            send = score_vectors_on_device(recs)
            recv = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=device)  # rank-major concat
            work = dist.all_gather_into_tensor(recv, send, async_op=True)
            state["pending_gather"] = (work, recv, recs["sad_h"].copy())

    def run(steps: int, timing: bool):
        if args.detector == "hash":
            for _ in range(steps):
                ms = wl.run_hash_step()
                if timing:
                    kernel_ms.append(ms)
            return
        wl.submit()
        for _ in range(steps - 1):
            wl.submit()
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
    algo_bytes = wl.algorithmic_bytes()
    achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9

    out = None
    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                key = f"{args.detector}_{args.res}_{args.dist}_{n}" + (f"_downscale_{args.downscale}" if args.downscale else "")
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        headline = (args.detector, args.res, args.downscale) == ("content", "1080p", None)
        what = f"detector set '{args.detector}'" + (f" behind the reference's downscale ({args.downscale}: {W}x{H} -> {wl.sw}x{wl.sh})"
                                                     if args.downscale else "")
        out = {
            "metric": "frames/sec (1080p ContentDetector)" if headline else
                      f"frames/sec ({args.res} {args.detector}" + (f", downscale {args.downscale})" if args.downscale else ")"),
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
                             f"(BASELINE.json configs[1]); distribution {args.dist}") if headline else
                            f"{what} on {n} x {W}x{H} BGR frames per GPU, device-resident; distribution {args.dist}",
                "frames_per_gpu": n, "height": H, "width": W, "distribution": args.dist,
                "parallelism": f"clips sharded over {world} GPU(s), RCCL all-gather of score vectors (device-resident records)"
                               if use_dist else "1 GPU",
                "pipeline_depth": 1 if args.detector == "hash" else 2,
                "exchange": (("host copy of the records (" + state["no_device_view"] + ")") if state.get("no_device_view") else
                             "score vectors sliced out of the device-resident records (psd_last_records_device)") if use_dist else None,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": wl.kernel_name(),
                "avg_launch_ms": round(avg_kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "limiter": "package power: the HSV pass holds the 1400 W cap and the shader clock falls to 1.9-2.1 GHz "
                           "(profiles/r02_a_power_and_clock_by_build.txt)" if headline else None,
            },
            "cuts_found": len(wl.state["cuts"]),
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
                                    if np.array_equal(wl.state["thumbs"][: len(sample)], ref) else "MISMATCH vs oracle")
        elif not args.no_cpu_baseline and world == 1 and not args.downscale and args.detector != "edges":
            sample = batch[: args.cpu_sample].cpu().numpy()
            cb = cpu_baseline(sample, wl.flags & 7, os.cpu_count() or 1)
            ref = cb.pop("_records")
            got = wl.state["recs"][: args.cpu_sample]
            same = all(np.array_equal(got[f], ref[f]) for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist"))
            out["cpu_baseline"] = cb
            out["parity_sample"] = "records of the first %d frames identical to the oracle" % args.cpu_sample if same \
                else "MISMATCH vs oracle"
        else:
            out["cpu_baseline"] = None
        if on_gpu and world == 1 and not use_dist and not args.no_secondary and headline:
            out["secondary"] = secondary_runs(eng, batch, E, epilogue, device, min(args.secondary_frames, n))
    eng.close()
    if use_dist:
        import torch.distributed as dist

        dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
