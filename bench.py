"""bench.py -- frames/sec of the ContentDetector hot path on device-resident 1080p batches.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 4096] [--dist U|S|K]
                    [--detector content|hist|all|hash|edges] [--res 1080p|4k] [--downscale auto|F] [--no-secondary]
                    [--workload headline|corpus|bbc]

Workload (BASELINE.json configs[1]): ContentDetector(threshold=27) over a synthetic
1920x1080 BGR batch of 4096 frames per GPU, resident in HBM when the timed region starts.
One "step" = one full pass of the hot path over the batch: the HIP scoring kernel,
the device->host copy of the per-frame records, and the native decision epilogue
(content_val + FlashFilter) that yields the cut list.  Steps are pipelined two deep
(the next kernel runs while the previous step's records are turned into cuts).

--downscale auto puts the reference's default downscale in front (SceneManager.auto_downscale,
scene_manager.py:110,123-140,666-678: cv2.resize to about 256 pixels width, INTER_LINEAR); the roofline then counts the
source rows that carry taps (2 per destination row), which is all that pipeline has to read.

N > 1: one rank per GPU.  Either the caller launches the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / WORLD_SIZE in the environment) or `python bench.py --gpus N` on its own starts them
(it re-executes itself under torch.distributed.run on 127.0.0.1).  Every rank scores its own batch (independent clips:
weak scaling, no data-path collective) and the per-frame score vectors are all-gathered over RCCL each step straight
from the device-resident records, so every rank could run the epilogue for all clips.

--workload corpus  BASELINE.json configs[4]: a mixed 1080p / 4K corpus of shot-like clips, all four detectors from one
                   fused pass per resolution (clips packed back to back in one device batch), sharded by clip, the
                   per-frame records all-gathered over RCCL, decisions on every rank.  Every rank holds its share of the
                   8-GPU corpus (24 x 1080p x 2048 + 8 x 4K x 512 frames => 3 + 1 clips per GPU): weak scaling.
--workload bbc     BASELINE.json configs[3]: AdaptiveDetector(window_width=2, min_content_val=15) over the 11-clip
                   stand-in for the BBC set (640x360, generated on the device), sharded by clip: strong scaling.

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
    elif dist == "T":  # shots with OBJECTS: the smooth shots of S plus flat rectangles and a bright diagonal band that drift from
        # frame to frame -- sharp edges for Canny (a few per cent of the pixels), between S (none) and U (every pixel)
        shot = 64
        ys = torch.arange(h, device=device).view(h, 1)
        xs = torch.arange(w, device=device).view(1, w)
        for s0 in range(0, n, shot):
            grid = torch.rand((1, 3, 9, 16), device=device, generator=g) * 255.0
            base = torch.nn.functional.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
            rects = torch.rand((12, 4), device=device, generator=g).cpu().tolist()
            cols = (torch.rand((12, 3), device=device, generator=g) * 255.0)
            for i in range(s0, min(n, s0 + shot)):
                k = i - s0
                img = base + torch.randn((h, w, 3), device=device, generator=g) * 2.0
                for (ry, rx, rh, rw), col in zip(rects, cols):
                    y0 = int(ry * (h - h // 4)) + k
                    x0 = int(rx * (w - w // 4)) + 2 * k
                    y1, x1 = min(h, y0 + h // 16 + int(rh * h // 6)), min(w, x0 + w // 16 + int(rw * w // 6))
                    img[y0:y1, x0:x1] = col
                band = (ys - (xs * h) // w - 3 * k).abs() < max(2, h // 90)
                img = torch.where(band.unsqueeze(2), img + 80.0, img)
                x[i] = img.round().clamp(0, 255).to(torch.uint8)
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


class SmiPoll:
    """Socket power and shader clock of the GPU while the timed steps run (amdgpu hwmon sysfs, 50 Hz, a daemon thread that only
    reads two small files): what the box was doing goes INTO the record, so that a line read on a slow box can be told from a
    regression (round-5 review: the driver's box read 4 - 6 % under the builder's, and nothing in BENCH_r05 said why).
    The HSV pass sits at the package power cap: its clock is what the firmware leaves it."""

    def __init__(self, device_index: int = 0):
        import glob

        def first(pattern):
            for path in sorted(glob.glob(pattern)):
                try:
                    open(path).read()
                    return path
                except OSError:
                    pass
            return None

        # The sysfs node of the device THIS process computes on, by its PCI address (a container may see the hwmon entries of other
        # tenants' GPUs of the node -- the first one that answers read 239 W / 105 MHz through eight seconds of this kernel,
        # tools/probe_power_ramp.py): no address or no readable node -> no record, rather than somebody else's GPU.
        pick = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            if glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/" % bdf):
                pick = "/sys/bus/pci/devices/%s/hwmon/hwmon*/" % bdf
            self.bdf = bdf
        except Exception:  # noqa: BLE001
            self.bdf = None
        if pick is None:
            self.pw = self.fq = self.cap = self.dpm = None
            self.samples, self._stop, self._thread = [], False, None
            return
        self.pw = first(pick + "power1_input") or first(pick + "power1_average")     # (instantaneous where the driver has it)
        self.fq = first(pick + "freq1_input")
        self.cap = first(pick + "power1_cap")
        # no hwmon clock on this driver: the DPM table marks the current shader-clock level with '*'
        dev = pick.split("/hwmon/")[0]
        self.dpm = None if self.fq else first(dev + "/pp_dpm_sclk")
        self.samples, self._stop, self._thread = [], False, None

    def _run(self):
        import re

        while not self._stop:
            row = {}
            try:
                if self.pw:
                    row["w"] = int(open(self.pw).read()) / 1e6
                if self.fq:
                    row["mhz"] = int(open(self.fq).read()) / 1e6
                elif self.dpm:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", open(self.dpm).read())
                    if m:
                        row["mhz"] = float(m.group(1))
            except (OSError, ValueError):
                pass
            if row:
                self.samples.append(row)
            time.sleep(0.02)

    def __enter__(self):
        import threading

        if self.pw or self.fq or self.dpm:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def summary(self) -> dict | None:
        if not self.samples:
            return None
        # (the sensors average over hundreds of milliseconds: the steady state is the second half of the samples)
        half = self.samples[len(self.samples) // 2:]
        ws = sorted(r["w"] for r in half if "w" in r)
        fs = sorted(r["mhz"] for r in half if "mhz" in r)
        out = {"samples": len(self.samples), "pci": self.bdf,
               "source": "amdgpu sysfs of this device (hwmon power, hwmon / DPM shader clock), 50 Hz, second half of the probe"}
        if ws:
            out.update(socket_power_w_median=round(ws[len(ws) // 2], 1), socket_power_w_max=round(ws[-1], 1))
        if fs:
            out.update(shader_clock_mhz_median=round(fs[len(fs) // 2]), shader_clock_mhz_min=round(fs[0]), shader_clock_mhz_max=round(fs[-1]))
        if self.cap:
            try:
                out["power_cap_w"] = int(open(self.cap).read()) / 1e6
            except (OSError, ValueError):
                pass
        return out


def parity_runs(n: int, k: int, walk: int = 0, frame_bytes: int = 0) -> list[tuple[int, int]]:
    """Frame ranges [a, b) a parity sample looks at: the first, the middle and the last k frames of the batch, both sides
    of every boundary between two time walks of the launch (`walk` frames each: psd_last_walk_geometry -- a workgroup starts
    from a re-read halo frame there) and both sides of the 2^32 / 2^33 / 2^34 / 2^35 byte offsets of the batch.  Merged and
    sorted."""
    want = set(range(0, min(n, k))) | set(range(max(0, n - k), n))
    mid = max(0, n // 2 - k // 2)
    want |= set(range(mid, min(n, mid + k)))
    if walk > 0:
        for b in range(walk, n, walk):
            want |= {b - 1, b}
    if frame_bytes > 0:
        for e in (32, 33, 34, 35):
            f = (1 << e) // frame_bytes
            if 1 <= f < n - 1:
                want |= {f - 1, f, f + 1}
    idx = sorted(want)
    runs, a = [], idx[0]
    for i, j in zip(idx, idx[1:] + [None]):
        if j != i + 1:
            runs.append((a, i + 1))
            a = j
    return runs


def describe_runs(runs: list[tuple[int, int]], walk: int = 0) -> str:
    total = sum(b - a for a, b in runs)
    head = ", ".join(f"{a}-{b - 1}" for a, b in runs[:3])
    tail = ", ".join(f"{a}-{b - 1}" for a, b in runs[-2:]) if len(runs) > 3 else ""
    s = f"{total} frames in {len(runs)} ranges ({head}" + (f", ..., {tail}" if tail else "") + ")"
    return s + (f", both sides of every {walk}-frame walk boundary" if walk > 0 else "")


def oracle_records_at(batch, runs, flags: int, threads: int, transform=None, edges: bool = False) -> list:
    """Oracle records of the frames in `runs` (each range scored with the frame in front of it as its predecessor)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import lib as orc
    from oracle.detectors_np import score_batch as oracle_score

    def work(r):
        a, b = r
        host = batch[max(a - 1, 0):b].cpu().numpy()
        if transform is not None:
            host = transform(host)
        prev, frames = (None, host) if a == 0 else (host[0], host[1:])
        return oracle_score(frames, prev, edges=True) if edges else orc.score_batch(frames, prev, flags=flags & 7)

    with ThreadPoolExecutor(max(1, min(threads, len(runs)))) as ex:
        return list(ex.map(work, runs))


def cpu_baseline(sample: np.ndarray, flags: int, threads: int, repeats: int = 3, model_frames: int = 256) -> dict:
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
    rates = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(work, bounds))
        rates.append(n / (time.perf_counter() - t0))
    out = {"value": round(float(np.median(rates)), 2), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{n} frames of the same batch through oracle/cv2_restate.c (gcc -O3), {threads} threads, median of "
                     f"{len(rates)} repeats ({min(rates):.0f}-{max(rates):.0f}); single thread: {single:.2f} frames/s; "
                     "real OpenCV is not installed",
           "single_thread_frames_per_s": round(single, 2), "repeats": [round(r, 1) for r in rates],
           "_records": np.concatenate(parts)}
    if flags & 1 and model_frames > 0:
        shim = os.path.join(ROOT, "oracle", "cv2_shim")
        if shim not in sys.path:
            sys.path.append(shim)
        from oracle.reference_loop import time_models

        runs = []
        for _ in range(max(1, repeats)):
            m = time_models(sample[: min(n, model_frames)])
            m.pop("_scores")
            runs.append(m)
        m = sorted(runs, key=lambda r: r["python_loop_frames_per_s"])[len(runs) // 2]
        out["reference_model"] = {
            "what": "ContentDetector.process_frame as the reference runs it (scene_manager.py:578-585): one Python process, "
                    "one frame at a time, cv2.cvtColor + cv2.split (C restatement) then 3x _mean_pixel_distance (the "
                    "reference's numpy expression, content_detector.py:29-36); median of %d repeats" % len(runs),
            "frames_per_s": m["python_loop_frames_per_s"], "cores": 1, "frames": m["python_loop_frames"],
            "numpy_half_only_frames_per_s": m["numpy_half_frames_per_s"],
            "repeats_frames_per_s": [r["python_loop_frames_per_s"] for r in runs],
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
        self.state = {"cuts": [], "recs": None, "bits": None}
        self.walk = 0        # frames per time walk of this workload's launch (0: the kernel does not walk / not known)

    # bytes one launch has to read: every BGR byte once; with the edge term 5 B/px; behind the default
    # downscale the source rows that carry taps (two per destination row, never more than all rows)
    def algorithmic_bytes(self) -> int:
        if self.downscale:
            rows = min(self.h, 2 * self.sh)
            return self.n * rows * self.w * 3
        per_px = 5 if self.detector == "edges" else ALGO_BYTES_PER_PX
        return self.n * self.h * self.w * per_px

    def kernel_name(self) -> str:
        if self.downscale and self.detector == "edges":
            return ("psd::resize_walk_kernel<VOUT> (fused downscale + HSV term; of the resized frame only its V plane and V histogram leave the CU) "
                    "+ the edge pipeline (Sobel / NMS, hysteresis, dilation + XOR) on the V planes")
        if self.downscale:
            return "psd::resize_walk_kernel"
        return {"content": "psd::score_frames_dma_kernel", "all": "psd::score_frames_dma_kernel", "hist": "psd::luma_hist_kernel",
                "hash": "psd::gray_area_dma_kernel + psd::hash_bits_kernel", "edges": "psd::score_frames_dma_kernel<V mode> + psd::sobel_nms_bits_kernel (edge pipeline)"}[self.detector]

    def submit(self):
        stream = self.stream
        alt = getattr(self, "alt_streams", None)
        if alt:      # --alt-streams (experiment): consecutive submissions on two streams, so that one launch's tail and the next one's ramp overlap
            self._k = getattr(self, "_k", 0) + 1
            stream = alt[self._k & 1]
        if self.downscale:
            self.eng.submit_device_downscaled(self.ptr, self.n, self.h, self.w, self.sh, self.sw, flags=self.flags, stream=stream)
        elif stream is not None:
            self.eng.submit_device(self.ptr, self.n, self.h, self.w, flags=self.flags, stream=stream)
        else:
            self.eng.submit_device(self.ptr, self.n, self.h, self.w, flags=self.flags)
        if not self.walk and (self.flags & 1) and hasattr(self.eng, "last_walk_geometry"):
            self.walk = int(self.eng.last_walk_geometry()[0])

    def finish(self) -> float:
        """Collect one submission and turn its records into cut lists; returns the kernel time (ms)."""
        # without the histogram term only the five sums of a record come back (40 instead of 1064 bytes per frame)
        recs = self.eng.collect(self.n, sums_only=not (self.flags & self.E.SCORE_LUMA_HIST))
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

    # HashDetector with the reference's defaults (hash_detector.py:68-77: size = 16, lowpass = 2): 32 x 32 grey thumbnails, the
    # 16 x 16 lowest DCT frequencies, 256 hash bits per frame.  (Rounds 3-5 timed size = 8: 16 x 16 thumbnails, 64 bits.)
    HASH_SIZE, HASH_LOWPASS = 16, 2

    def run_hash_steps(self, k: int, sink: list) -> None:
        """k HashDetector steps.  Round 6: thumbnails, DCT, median and bits are ONE device call (psd_hash_bits_device: 256 bytes
        per frame come back), then the native decisions; until round 5 the step ended in the DCT on the host (1.1 ms per 4096
        frames on 16 threads, hidden behind the next step's kernel by a second thread: 865 k frames/s, 744 k without that)."""
        im = self.HASH_SIZE * self.HASH_LOWPASS
        for _ in range(k):
            bits = self.eng.hash_bits_device(self.ptr, self.n, self.h, self.w, im, self.HASH_SIZE)
            sink.append(self.eng.last_kernel_ms()[0])
            self.state["cuts"] = self.ep.hash_cuts(bits, 25.0, threshold=0.35, min_scene_len=15)[0]
            self.state["bits"] = bits


def parity_sample(wl: Workload, k: int) -> str:
    """The workload's last step against the CPU oracle (the checker, outside every timed region): the first, middle and
    last k frames of the batch plus both sides of every boundary between two time walks of the launch (`parity_runs`)."""
    from oracle import lib as orc

    threads = os.cpu_count() or 1
    if wl.detector == "hash":
        k = min(k, wl.n)
        im, hs = wl.HASH_SIZE * wl.HASH_LOWPASS, wl.HASH_SIZE
        bad = []
        for a in sorted({0, max(0, wl.n // 2 - k // 2), wl.n - k}):      # the first, the middle and the last k frames
            frames = wl.batch[a:a + k].cpu().numpy()
            want_thumbs = orc.hash_thumbs(frames, im)
            _, thumbs = wl.eng.hash_bits_device(wl.ptr + a * wl.h * wl.w * 3, k, wl.h, wl.w, im, hs, want_thumbs=True)
            if not np.array_equal(thumbs, want_thumbs):
                bad.append("thumbnails of frames %d-%d" % (a, a + k - 1))
            if not np.array_equal(wl.state["bits"][a:a + k], orc.hash_bits(want_thumbs, hs).reshape(k, -1)):
                bad.append("hash bits of frames %d-%d" % (a, a + k - 1))
        what = "%d x %d thumbnails and %d hash bits of the first, middle and last %d frames" % (im, im, hs * hs, k)
        return (what + " identical to the oracle") if not bad else "MISMATCH vs oracle: " + ", ".join(bad)
    edges = wl.detector == "edges"
    # (the edge term's oracle is two orders of magnitude slower than the sums: walk boundaries only for the cheap terms)
    runs = parity_runs(wl.n, k, 0 if edges else wl.walk, 0 if wl.downscale else wl.h * wl.w * 3)
    transform = None
    if wl.downscale:
        shim = os.path.join(ROOT, "oracle", "cv2_shim")
        if shim not in sys.path:
            sys.path.append(shim)
        import cv2  # the oracle's shim: cv2.resize restated (scene_manager.py:670-678)

        def transform(host):
            return np.stack([cv2.resize(f, (wl.sw, wl.sh)) for f in host])
    wants = oracle_records_at(wl.batch, runs, wl.flags, threads, transform, edges)
    fields = (["sad_h", "sad_s", "sad_v"] if wl.flags & 1 else []) + (["hist"] if wl.flags & 2 else []) + \
             (["byte_sum"] if wl.flags & 4 else []) + (["edge_xor"] if wl.flags & 8 else [])
    got = wl.state["recs"]
    bad = sorted({f for (a, b), want in zip(runs, wants) for f in fields if f in got.dtype.names and not np.array_equal(got[f][a:b], want[f])})
    what = "records (" + ", ".join(fields) + ") of " + describe_runs(runs, 0 if edges else wl.walk)
    return (what + " identical to the oracle") if not bad else "MISMATCH vs oracle in %s: %s" % (bad, what)


def quick_measure(wl: Workload, steps: int = 5, warmup: int = 2, parity_frames: int = 16) -> dict:
    """Short single-GPU measurement of a secondary workload: same step definition, two steps in flight."""
    def run(k, sink):
        if wl.detector == "hash":
            wl.run_hash_steps(k, sink)
            return
        wl.submit()
        for _ in range(k - 1):
            wl.submit()
            sink.append(wl.finish())
        sink.append(wl.finish())

    # Warm up by TIME as well as by count: every secondary follows seconds of CPU work (the previous line's parity sample) with the
    # GPU idle, and a GPU that wakes up reclocks under load some tens of milliseconds in -- one stall of 5 ... 37 ms
    # (tools/experiments_r05/probe21.py: steps of 1.15 ms, the fourth ten of them took 48.9 ms).  Inside a 20-step window of a 1 ms
    # kernel that read as 1.36 ms per step in this list against 1.195 ms for the same workload run on its own.
    t_warm = time.perf_counter()
    run(warmup, [])
    torch.cuda.synchronize()
    while time.perf_counter() - t_warm < 0.12:
        run(max(2, warmup), [])
        torch.cuda.synchronize()
    ms: list[float] = []
    t0 = time.perf_counter()
    run(steps, ms)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(ms))
    achieved = wl.algorithmic_bytes() / (k_ms * 1e-3) / 1e9
    out = {"value": round(wl.n * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 4),
           "avg_launch_ms": round(k_ms, 4), "achieved_GBps": round(achieved, 1), "frac_of_8TBps": round(achieved / HBM_PEAK_GBS, 4),
           "kernel": wl.kernel_name(), "cuts_found": len(wl.state["cuts"])}
    if parity_frames > 0:
        try:
            out["parity_sample"] = parity_sample(wl, parity_frames)
        except Exception as ex:  # noqa: BLE001
            out["parity_sample"] = "not checked: %s: %s" % (type(ex).__name__, ex)
    return out


def secondary_runs(eng, batch, E, epilogue, device, frames_small: int, args) -> dict:
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
            lambda: quick_measure(Workload(eng, batch, "all", None, epilogue, E), steps=10, warmup=3))
    attempt("default_pipeline_downscale_auto", f"ContentDetector behind SceneManager's default downscale ({w}x{h} -> 256 wide), {n} frames; "
            "roofline counts the source rows that carry taps",
            lambda: quick_measure(Workload(eng, batch, "content", "auto", epilogue, E), steps=20, warmup=5))   # (1.2 ms per step)
    attempt("default_pipeline_downscale_auto_all_four", f"all four detectors behind SceneManager's default downscale ({w}x{h} -> 256 wide), {n} frames: "
            "resize + HSV + luma histogram + byte sum in ONE kernel, the small frame never exists in memory; roofline counts the source rows "
            "that carry taps", lambda: quick_measure(Workload(eng, batch, "all", "auto", epilogue, E), steps=20, warmup=5))
    def run_ds_edges():
        # the reference's default pipeline WITH a StatsManager (content_detector.py:158: the edge term is computed whenever one is
        # attached): resize to 256 wide into the engine's buffer, then HSV + Canny / dilate / XOR on the small frames, k = 5
        bt = make_batch(frames_small, "T", 20250921, device, h, w)
        r = quick_measure(Workload(eng, bt, "edges", "auto", epilogue, E), steps=10, warmup=3)
        del bt
        return r
    attempt("default_pipeline_downscale_auto_edges", f"ContentDetector with the edge term (weights 1,1,1,1: what a StatsManager makes the reference "
            f"compute) behind SceneManager's default downscale ({w}x{h} -> 256 wide), {frames_small} frames with objects; roofline counts "
            "the source rows that carry taps", run_ds_edges)
    attempt("hash_detector_1080p", f"HashDetector with the reference's defaults (32 x 32 thumbnails, 256 bits): thumbnail kernel + DCT / median / bits "
            f"on the device (psd_hash_bits_device), {n} x {w}x{h}",
            lambda: quick_measure(Workload(eng, batch, "hash", None, epilogue, E), steps=5, warmup=2))
    for dist, label in (("S", "shot-like content (64-frame shots, hard cuts)"), ("K", "constant frames (one histogram bin per frame)")):
        def run_dist(dist=dist):
            # shot-like content at the headline's batch length (the fraction of the roofline moves with the length of a
            # launch, not with the content: like for like), constant frames at the short one
            b = make_batch(n if dist == "S" else frames_small, dist, 20250921, device, h, w)
            r = quick_measure(Workload(eng, b, "content", None, epilogue, E))
            if dist == "S":
                r["edges_weights_1111"] = quick_measure(Workload(eng, b, "edges", None, epilogue, E), steps=3, warmup=1)
                # the same with sharp-edged objects in the frames (S holds next to no Canny edges, U nothing else)
                # (at the secondary batch length, 2048 frames: like the headline, the term's fraction moves with the length of a
                #  submission -- 1024-frame submissions read 0.02 lower)
                bt = make_batch(frames_small, "T", 20250921, device, h, w)
                r["edges_weights_1111_objects"] = quick_measure(Workload(eng, bt, "edges", None, epilogue, E), steps=5, warmup=2)
                del bt
                r["all_four_fused"] = quick_measure(Workload(eng, b, "all", None, epilogue, E))
            else:
                r["histogram_threshold"] = quick_measure(Workload(eng, b, "hist", None, epilogue, E))
            del b
            return r
        attempt(f"content_1080p_{dist}", f"ContentDetector, {n if dist == 'S' else frames_small} x {w}x{h}, {label}", run_dist)

    def run_4k():
        b = make_batch(frames_small, "U", 20250921, device, 2160, 3840)
        r = quick_measure(Workload(eng, b, "hist", None, epilogue, E))
        r["content_detector"] = quick_measure(Workload(eng, b, "content", None, epilogue, E))
        del b
        return r
    attempt("histogram_threshold_4k", f"BASELINE configs[2]: HistogramDetector + ThresholdDetector, {frames_small} x 3840x2160, uniform bytes", run_4k)
    torch.cuda.empty_cache()

    def run_flow_secondary(kind):
        # The two flows at the length BASELINE.md quotes them at (round-5 review: the driver-visible line ran them shortened and read
        # other fractions than the table): one GPU's share of the corpus, 3 x 2048 x 1080p + 512 x 4K, and the 11-clip BBC stand-in
        # with 6000+ frames per clip -- 51 GB each, generated once and scored through BOTH pipelines: the reference's default (every
        # frame behind SceneManager's auto-downscale: the primary numbers of the line) and full resolution (`full_resolution`).
        import copy

        a = copy.copy(args)
        # (steps / warmup as `bench.py --workload <kind>` on its own runs them: with one warm-up pass the three timed ones still paid for the first use of
        #  the decision threads and the record mirrors -- the corpus flow read 4.4 ms per pass here and 3.5 ms on its own, same kernels)
        a.workload, a.corpus_frames, a.bbc_frames, a.steps, a.warmup, a.no_cpu_baseline = kind, args.flow_corpus_frames, args.flow_bbc_frames, 6, 3, False
        a.height = a.width = 0
        a.cpu_sample = 256
        a.flow_pipeline = "default"
        fw = FlowWorkload(kind, eng, device, 0, 1, a)
        keys = ("value", "unit", "ms_per_step", "roofline", "pipeline", "cuts_found", "ground_truth", "parity_sample", "cpu_baseline")

        def brief(r):
            return {k: r[k] for k in keys} | {"frames_1080p_equivalent_per_s": r["config"]["frames_1080p_equivalent_per_s"],
                                              "frames_total": r["config"]["frames_total"]}

        out_ = brief(run_flow(a, eng, device, 1, 0, device.index or 0, False, True, fw=fw))
        fw.set_pipeline("full")
        out_["full_resolution"] = brief(run_flow(a, eng, device, 1, 0, device.index or 0, False, True, fw=fw))
        del fw
        torch.cuda.empty_cache()
        return out_
    def run_host_fed():
        # PCIe-inclusive and never `value`: SceneManager.detect_scenes over frames in pageable host memory, the reference's
        # default pipeline (auto downscale to 256 wide, ContentDetector).  The decode thread uploads only the rows the
        # downscale reads (psd_upload_rows) while the previous batch is scored.
        import pyscenedetect_amd as psd
        from pyscenedetect_amd.scene_manager import compute_downscale_factor

        n_fed = 768
        b = make_batch(n_fed, "S", 20250921, device, h, w)
        host = b.cpu().numpy()
        del b
        factor = compute_downscale_factor(w)
        dw, dh = max(1, round(w / factor)), max(1, round(h / factor))
        rows = eng.downscale_source_rows(h, w, dh, dw, 1)

        def run(engine, frames, with_stats=True):
            stats = psd.StatsManager() if with_stats else None
            sm = psd.SceneManager(stats, engine=engine)
            sm.add_detector(psd.ContentDetector(engine=engine))
            sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
            vals = [stats.get_metrics(i, ["content_val"])[0] for i in range(1, len(frames))] if with_stats else None
            return [c.frame_num for c in sm.get_cut_list(show_warning=False)], vals

        # The timed run is the reference's DEFAULT pipeline: `detect()` attaches a StatsManager only when a stats file is asked
        # for (scenedetect/__init__.py:208-210), so the default has none -- no per-frame metric dictionaries, and no edge term
        # (content_detector.py:158).  The same run with a StatsManager (per-frame metrics kept, edge term on) is reported beside it.
        class Cycled:
            """The stored frames `times` times over as one clip (any indexable of frames is a source): a timed run of 768 frames
            lasts 28 ms, in which the start of the feeder's threads and the first touch of its staging segments showed as 20-30 k
            frames/s from one invocation to the next; 3072 frames last 0.1 s."""
            def __init__(self, frames, times):
                self.frames, self.n, self.times = frames, len(frames), times
            def __len__(self):
                return self.n * self.times
            def __getitem__(self, i):
                return self.frames[i % self.n]

        clip = Cycled(host, 4)
        run(eng, host[:64], False)
        best, cuts = 1e9, None
        for _ in range(3):
            t0 = time.perf_counter()
            cuts, _ = run(eng, clip, False)
            best = min(best, time.perf_counter() - t0)
        best_stats = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            cuts_stats, _vals = run(eng, clip, True)
            best_stats = min(best_stats, time.perf_counter() - t0)
        # `value` keeps the definition of rounds 1-4 -- WITH a StatsManager (per-frame metrics kept, the edge term computed) -- so that the
        # line compares across rounds (round 5 had silently made the no-StatsManager run the value: ADVICE r05); the reference's default
        # `detect()` (no StatsManager, scenedetect/__init__.py:208-210) is reported under its own key
        r = {"value": round(len(clip) / best_stats, 1), "unit": "frames/s", "pcie_inclusive": True, "frames": len(clip),
             "distinct_frames": len(host),
             "definition": "SceneManager.detect_scenes WITH a StatsManager over %d frames (%d distinct ones, cycled), thread on the GPU's NUMA "
                           "node; best of 2" % (len(clip), len(host)),
             "with_stats_manager_frames_per_s": round(len(clip) / best_stats, 1),
             "without_stats_manager_frames_per_s": round(len(clip) / best, 1),
             "source_rows_uploaded_per_frame": int(len(rows)), "source_rows_per_frame": h,
             "frames_per_upload_call": psd.scene_manager._DeviceFeeder.FEED_BATCH, "cuts_found": len(cuts),
             "host_to_device_GBps": round(len(rows) * w * 3 * len(clip) / best / 1e9, 2)}
        if cuts_stats != cuts:
            r["parity_sample"] = "MISMATCH: cut lists with and without a StatsManager differ"
            return r
        try:
            from oracle.detectors_np import OracleEngine

            shim = os.path.join(ROOT, "oracle", "cv2_shim")
            if shim not in sys.path:
                sys.path.append(shim)
            k = 48
            same = run(eng, host[:k]) == run(OracleEngine(), host[:k])
            r["parity_sample"] = ("cuts and content_val of the first %d frames identical to the oracle engine behind the same SceneManager" % k
                                  if same else "MISMATCH vs the oracle engine (first %d frames)" % k)
        except Exception as ex:  # noqa: BLE001
            r["parity_sample"] = "not checked: %s: %s" % (type(ex).__name__, ex)
        return r
    def run_per_frame():
        # The per-frame plug-in API north_star keeps (detector.py:48-60: SceneDetector.process_frame(timecode, frame_img)),
        # one frame per call from pageable host memory, full resolution, beside the reference's own execution model
        # (`cpu_baseline.reference_model`: the same loop over cv2 / numpy on one core).  PCIe-inclusive, never `value`.
        import pyscenedetect_amd as psd
        from pyscenedetect_amd.timecode import FrameTimecode

        b = make_batch(160, "S", 20250921, device, h, w)
        host = b.cpu().numpy()
        del b
        tcs = [FrameTimecode(i, 25.0) for i in range(len(host))]

        def mirror(engine, frames):
            det = psd.ContentDetector(engine=engine)
            cuts = []
            for i, f in enumerate(frames):
                cuts += det.process_frame(tcs[i], f)
            return [c.frame_num for c in cuts], det

        mirror(eng, host[:8])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            cuts, _det = mirror(eng, host)
            best = min(best, time.perf_counter() - t0)
        r = {"value": round(len(host) / best, 1), "unit": "frames/s", "pcie_inclusive": True, "frames": len(host), "cuts_found": len(cuts),
             "api": "pyscenedetect_amd.ContentDetector.process_frame(timecode, frame): one upload (the previous frame stays in HBM), "
                    "psd_score_batch_device(n=1), decision", "us_per_frame": round(best / len(host) * 1e6, 1)}
        # the binding a reference maintainer adds scores one frame per call from the reference's own per-frame
        # loop (integration/scenedetect_amd.py: _calculate_frame_score -> FramePair.score_next); timed here exactly as that
        # function issues it (the reference package itself is not on this box)
        sys.path.insert(0, os.path.join(ROOT, "integration"))
        import scenedetect_amd as B

        bind = B.Binding(os.path.join(ROOT, "pyscenedetect_amd", "libpsd_hip.so"), device.index or 0)

        def binding_loop(frames):
            pair, sads = bind.frame_pair(), []
            for f in frames:
                rec, _had_prev = pair.score_next(f, B.HSV_SAD)
                sads.append((rec.sad_h, rec.sad_s, rec.sad_v))
            pair.release()
            return sads

        binding_loop(host[:4])
        best_b = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            sads = binding_loop(host)
            best_b = min(best_b, time.perf_counter() - t0)
        bind.close()
        r["reference_binding"] = {"value": round(len(host) / best_b, 1), "unit": "frames/s", "pcie_inclusive": True,
                                  "us_per_frame": round(best_b / len(host) * 1e6, 1),
                                  "api": "integration/scenedetect_amd.py FramePair.score_next = psd_memcpy_h2d + psd_score_batch_device(n=1, d_prev): "
                                         "what the patched reference ContentDetector._calculate_frame_score calls per frame (the previous frame "
                                         "stays in HBM: one upload per frame)"}
        try:
            from oracle import lib as orc
            from oracle.detectors_np import OracleEngine

            k = 48
            want = orc.score_batch(host[:k])
            same = mirror(eng, host[:k])[0] == mirror(OracleEngine(), host[:k])[0] and \
                all(tuple(int(want[f][i]) for f in ("sad_h", "sad_s", "sad_v")) == tuple(int(v) for v in sads[i]) for i in range(k))
            r["parity_sample"] = ("process_frame cuts and the binding's sums of the first %d frames identical to the oracle" % k) if same \
                else "MISMATCH vs the oracle (first %d frames)" % k
        except Exception as ex:  # noqa: BLE001
            r["parity_sample"] = "not checked: %s: %s" % (type(ex).__name__, ex)
        return r
    def near_gpu(fn):
        # The two PCIe-inclusive lines run the way a deployment would be started (numactl --cpunodebind): the thread that decodes
        # -- here: that creates the frames -- and its children on the CPUs of the GPU's NUMA node (psd_cpus_near_device).  Left to
        # the OS the frames land on either socket by the invocation, and the host-fed line reads 24-31 k frames/s accordingly.
        def wrapped():
            cpus = eng.cpus_near_gpu() if hasattr(eng, "cpus_near_gpu") else []
            before = os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
            try:
                r = fn()
            finally:
                os.sched_setaffinity(0, before)
            r["cpu_affinity"] = ("the %d CPUs of the GPU's NUMA node" % len(cpus)) if cpus else "unchanged (one node, or nothing to steer)"
            return r
        return wrapped

    attempt("per_frame_api_1080p", f"PCIe-inclusive: the per-frame plug-in API (SceneDetector.process_frame) and the reference-side "
            f"binding of INTEGRATION.md B on 160 x {w}x{h} shot-like frames, one frame per call", near_gpu(run_per_frame))
    attempt("host_fed_default_pipeline", f"PCIe-inclusive: SceneManager.detect_scenes (auto downscale, ContentDetector, no StatsManager: the "
            f"reference's default) over 3072 x {w}x{h} shot-like frames (768 distinct ones, four times over) in pageable host memory; only the source "
            "rows that carry taps are uploaded, 16 frames per call", near_gpu(run_host_fed))
    attempt("corpus_mixed_1080p_4k_all_four", f"BASELINE configs[4], one GPU's share at full length: 3 x 1080p x {args.flow_corpus_frames} + 1 x 4K x "
            f"{max(1, args.flow_corpus_frames // 4)} shot-like frames, all four detectors, clips packed per resolution, every frame behind the "
            "reference's default downscale (= bench.py --workload corpus); `full_resolution`: the same clips without the resize",
            lambda: run_flow_secondary("corpus"))
    attempt("bbc_standin_adaptive", f"BASELINE configs[3] at full length: AdaptiveDetector over the 11-clip 640x360 stand-in, {args.flow_bbc_frames}+ "
            "frames per clip, every frame behind the reference's default downscale (= bench.py --workload bbc); `full_resolution`: without the resize",
            lambda: run_flow_secondary("bbc"))
    return out


ALL_FOUR = {"content": {}, "adaptive": {}, "hist": {}, "threshold": {}}          # reference constructor defaults
BBC_ADAPTIVE = {"adaptive": {"window_width": 2, "min_content_val": 15.0}}


class FlowWorkload:
    """BASELINE.json configs[3] / [4] as flows: a list of clips -> `corpus.detect_corpus` (pack by resolution, one fused
    launch per term per resolution, all-gather of the per-frame records over the process group, native decision
    epilogues on every rank) -> cut lists per clip and detector.  Every rank generates only the clips the greedy plan
    (`distributed.assign_clips`, the same plan detect_corpus uses) gives it."""

    def __init__(self, kind: str, eng, device, rank: int, world: int, args, small: bool = False):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import device_clips as DC
        from pyscenedetect_amd.distributed import assign_clips

        self.kind, self.eng, self.rank, self.world = kind, eng, rank, world
        # "default": every clip scored as scenedetect.detect(video, detector) scores it -- resized by SceneManager's auto_downscale
        # to about 256 pixels (scene_manager.py:110,123-140,666-678; what the reference's benchmark runs, benchmark/__main__.py:44-61);
        # "full": no resize (SceneManager.auto_downscale = False), the form rounds 2-5 timed
        self.set_pipeline(getattr(args, "flow_pipeline", "default"))
        if kind == "corpus":
            f = args.corpus_frames
            hw = [(args.height, args.width), (2 * args.height, 2 * args.width)] if args.height and args.width else [(1080, 1920), (2160, 3840)]
            # one GPU's share of the 8-GPU corpus (24 x 1080p x 2048 + 8 x 4K x 512): 3 + 1 clips; N ranks hold N shares
            self.specs = [(f, *hw[0])] * (3 * world) + [(max(1, f // 4), *hw[1])] * world
            self.detectors, self.scaling = ALL_FOUR, "weak"
            self.what = (f"mixed corpus, all four detectors: {3 * world} x {hw[0][1]}x{hw[0][0]} x {f} + {world} x {hw[1][1]}x{hw[1][0]} x "
                         f"{max(1, f // 4)} shot-like frames (3 + 1 clips per GPU: the per-GPU share of BASELINE.json configs[4] "
                         "at 8 GPUs), clips of one resolution packed into one device batch, sharded by clip, records all-gathered")
            shot_len = (40, 400)
        else:
            hw = (args.height, args.width) if args.height and args.width else (360, 640)
            self.specs = [(args.bbc_frames + 137 * i, *hw) for i in range(11)]
            self.detectors, self.scaling = BBC_ADAPTIVE, "strong"
            self.what = (f"AdaptiveDetector(window_width=2, min_content_val=15) on the 11-clip stand-in for the BBC set "
                         f"({hw[1]}x{hw[0]}, {sum(x[0] for x in self.specs)} frames, generated on the device; the real set "
                         "and a decoder are not in this image), sharded by clip (BASELINE.json configs[3])")
            shot_len = (40, 400)
        self.what_clips = self.what
        self.set_pipeline(self.pipeline)
        if small:
            shot_len = (8, 40)
        self.plan = assign_clips([n * h * w for n, h, w in self.specs], world)
        mine = self.plan[rank]
        built, truths = DC.make_packed_clips([self.specs[i] for i in mine], [5000 + i for i in mine], device, shot_len)
        self.clips = [DC.LazyClip(*sp) for sp in self.specs]
        self.truth = {}
        for i, c, t in zip(mine, built, truths):
            self.clips[i] = c
            self.truth[i] = t
        self.frames_total = sum(sp[0] for sp in self.specs)
        self.result = None

    def set_pipeline(self, pipeline: str) -> None:
        self.pipeline = pipeline
        self.downscale = "auto" if pipeline == "default" else None
        if hasattr(self, "what_clips"):
            self.what = self.what_clips + ("; every frame behind the reference's default downscale (SceneManager.auto_downscale, one factor per "
                                           "resolution: psd_score_segments_downscaled_device)" if self.downscale else
                                           "; FULL-resolution frames (auto_downscale off)")
        self.result = None

    def scored(self, h: int, w: int) -> tuple[float, int, int]:
        from pyscenedetect_amd.engine import downscale_size

        return downscale_size(h, w, self.downscale)

    def my_bytes(self) -> int:
        """Algorithmic bytes of this rank's clips: 3 B/px once; behind the downscale the source rows that carry taps
        (two per destination row, never more than all rows: DESIGN.md 4.4)."""
        total = 0
        for i in self.plan[self.rank]:
            n, h, w = self.specs[i]
            factor, dh, _ = self.scored(h, w)
            total += n * (min(h, 2 * dh) if factor > 1.0 else h) * w * 3
        return total

    def step(self) -> float:
        """One pass over the whole corpus; returns this rank's kernel time (ms, HIP events, summed over its launches)."""
        from pyscenedetect_amd.corpus import detect_corpus

        if hasattr(self.eng, "kernel_ms_acc"):
            self.eng.kernel_ms_acc = 0.0
        self.result = detect_corpus(self.eng, self.clips, 25.0, self.detectors, auto_downscale=self.downscale == "auto")
        return float(getattr(self.eng, "kernel_ms_acc", 0.0))

    def parity(self, budget_frames: int, threads: int) -> tuple[str, dict]:
        """This rank's first clip of every resolution, a prefix of `budget_frames` 1080p-equivalents each, through the CPU
        oracle: records and the cut lists decided from them must equal what the flow produced for that prefix."""
        from concurrent.futures import ThreadPoolExecutor

        from oracle import lib as orc
        from pyscenedetect_amd import corpus
        from pyscenedetect_amd.corpus import required_flags

        flags = required_flags(self.detectors)
        fields = (["sad_h", "sad_s", "sad_v"] if flags & 1 else []) + (["hist"] if flags & 2 else []) + (["byte_sum"] if flags & 4 else [])
        seen, notes, bad = set(), [], []
        t_cpu = n_cpu = 0.0
        for i in self.plan[self.rank]:
            n, h, w = self.specs[i]
            if (h, w) in seen:
                continue
            seen.add((h, w))
            k = int(max(2, min(n, budget_frames * (1080 * 1920) // (h * w))))
            frames = self.clips[i][:k].cpu().numpy()
            factor, dh, dw = self.scored(h, w)
            bounds = [(j * k // threads, (j + 1) * k // threads) for j in range(threads)]
            bounds = [b for b in bounds if b[1] > b[0]]

            resize = None
            if factor > 1.0:            # the oracle's cv2.resize (INTER_LINEAR) in front, like the reference's decode thread
                shim = os.path.join(ROOT, "oracle", "cv2_shim")
                if shim not in sys.path:
                    sys.path.append(shim)
                import cv2  # the oracle's shim: cv2.resize restated (scene_manager.py:670-678)

                def resize(host, dw=dw, dh=dh):
                    return np.stack([cv2.resize(f, (dw, dh)) for f in host])

            def oracle_part(r, resize=resize):
                part = frames[r[0] - 1 if r[0] else 0:r[1]]
                if resize is not None:
                    part = resize(part)
                return orc.score_batch(part[1:] if r[0] else part, part[0] if r[0] else None, flags=flags & 7)

            t0 = time.perf_counter()
            with ThreadPoolExecutor(threads) as ex:
                parts = list(ex.map(oracle_part, bounds))
            t_cpu += time.perf_counter() - t0
            n_cpu += k * (h * w) / (1080 * 1920)
            want = np.concatenate(parts)
            got_all = self._records_of(i)
            got = got_all[:k]
            if any(not np.array_equal(got[f], want[f]) for f in fields):
                bad.append(f"records of clip {i}")
            # ... and deep into the clip (it sits somewhere inside a packed batch): its middle and its last frames
            deep = [r for r in parity_runs(n, max(2, k // 4)) if r[0] >= k]
            for (a, b), w_ in zip(deep, oracle_records_at(self.clips[i], deep, flags, threads, resize)):
                if any(not np.array_equal(got_all[f][a:b], w_[f]) for f in fields):
                    bad.append(f"records of clip {i}, frames {a}-{b - 1}")
            want_cuts = corpus.decide(want, dh, dw, 25.0, self.detectors)
            if corpus.decide(got, dh, dw, 25.0, self.detectors) != want_cuts:
                bad.append(f"cuts of clip {i}")
            # ... and what the timed flow itself returned for this clip (every detector decides causally up to a short
            # look-ahead, so away from the end of the prefix the lists must agree)
            for name, cuts in want_cuts.items():
                if [c for c in self.result[i][name] if c < k - 32] != [c for c in cuts if c < k - 32]:
                    bad.append(f"{name} cuts the flow returned for clip {i}")
            notes.append(f"clip {i} ({w}x{h}" + (f" -> {dw}x{dh}" if factor > 1.0 else "") + f"): first {k} frames" +
                         ("".join(f", frames {a}-{b - 1}" for a, b in deep)))
        how = ("the reference pipeline's records and cut lists (cv2.resize to the auto-downscale size, then the detectors' arithmetic) "
               "identical to the oracle's: " if self.downscale else "records and cut lists of FULL-resolution frames identical to the oracle: ")
        msg = (how + "; ".join(notes)) if not bad else "MISMATCH vs oracle: " + ", ".join(bad)
        cpu = {"value": round(n_cpu / t_cpu, 2) if t_cpu > 0 else None, "unit": "1080p-equivalent frames/s", "cores": threads, "kind": "port",
               "sample": "the parity prefixes (" + "; ".join(notes) + f") through oracle/cv2_restate.c, {threads} threads, pixel work only"}
        return msg, cpu

    def _records_of(self, i: int):
        # the flow returns decisions; the records they came from are re-scored for the parity prefix (same engine calls)
        from pyscenedetect_amd import corpus

        return corpus.score_clip(self.eng, self.clips[i], corpus.required_flags(self.detectors), downscale=self.downscale)

    def truth_f1(self) -> dict | None:
        """Precision / recall / F1 of this rank's clips against the generator's ground truth, scored the way the reference's
        benchmark scores detectors (benchmark/evaluator.py: greedy 1-to-1 matching within a frame tolerance, counts summed
        over videos; tools/bbc_scoring.py restates it and passes the reference's own tests for it) at tolerances 0 and 1."""
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pathlib import Path

        from bbc_scoring import GroundTruth, Prediction, evaluate

        name = "adaptive" if "adaptive" in self.detectors else "content"
        preds = {Path(f"clip{i}"): Prediction(predicted_cuts=list(self.result[i][name]), ground_truth=GroundTruth(hard_cuts=list(t)), elapsed=0.0)
                 for i, t in self.truth.items()}
        if not preds:
            return None
        out = {"detector": name, "scope": "clips of rank 0", "scorer": "tools/bbc_scoring.py (the reference's benchmark/evaluator.py convention)"}
        for tol in (0, 1):
            hc = evaluate(preds, tol).hard_cuts
            if tol == 0:
                if not (hc.matched + hc.false_positives) or not (hc.matched + hc.missed):
                    return None
                out.update(precision=round(hc.precision, 4), recall=round(hc.recall, 4), f1=round(hc.f1, 4),
                           true_cuts=hc.matched + hc.missed, detected=hc.matched + hc.false_positives)
            else:
                out["f1_tolerance_1"] = round(hc.f1, 4)
        return out


def run_flow(args, eng, device, world, rank, local_rank, use_dist, on_gpu, fw=None) -> dict | None:
    """--workload corpus|bbc: W untimed passes, then K timed passes between barriers; MAX over ranks; rank 0 reports.
    `fw`: clips that exist already (the default run scores one set through both pipelines)."""
    if fw is None:
        fw = FlowWorkload(args.workload, eng, device, rank, world, args, small=not on_gpu)

    def barrier():
        if use_dist:
            import torch.distributed as dist

            dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        fw.step()
    barrier()
    kms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kms.append(fw.step())
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    k_ms = float(np.mean(kms)) if kms else float("nan")
    algo = fw.my_bytes()
    # HBM bytes per pass from the committed PMC count of exactly this flow (profiles/hbm_traffic.json), where there is one
    flow_traffic = (None, "not counted for this workload (same kernels as --detector all / content behind --downscale auto: profiles/hbm_traffic.json)")
    if args.workload == "bbc" and fw.downscale and world == 1 and args.bbc_frames == 6000 and not (args.height or args.width):
        try:
            with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
                ent = json.load(f)["bbc_standin_640x360_downscale_auto_r06"]
            flow_traffic = (ent["hbm_bytes_per_pass"], "profiles/hbm_traffic.json['bbc_standin_640x360_downscale_auto_r06']: %s -- a committed PMC "
                            "measurement of this flow, not counted in this run" % ent["source"])
        except Exception:  # noqa: BLE001
            pass
    achieved = algo / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None      # (the CPU dry run has no kernel clock)
    px_total = sum(n * h * w for n, h, w in fw.specs)
    out = {
        "metric": "frames/sec (mixed 1080p/4K corpus, all four detectors)" if args.workload == "corpus" else
                  "frames/sec (AdaptiveDetector, BBC stand-in, sharded by clip)",
        "value": round(fw.frames_total * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": fw.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": fw.what, "clips": len(fw.specs), "frames_total": fw.frames_total,
                   "frames_1080p_equivalent_per_s": round(px_total / (1080 * 1920) * args.steps / elapsed, 1),
                   "detectors": sorted(fw.detectors), "ranks_seen": world,
                   "parallelism": (f"clips sharded over {world} GPU(s) (greedy by frames x pixels), all-gather of the per-frame records "
                                   "(RCCL), decisions on every rank") if use_dist else "1 GPU",
                   "clips_of_rank0": fw.plan[0]},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": flow_traffic[0],
                     "traffic_source": flow_traffic[1],
                     "kernel": ("psd::resize_walk_kernel<SEG> (fused downscale + score; one launch per resolution, clips packed; rank 0's launches)"
                                if fw.downscale else "psd::score_frames_dma_kernel (one launch per resolution, clips packed; rank 0's launches)"),
                     "avg_launch_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": algo,
                     "note": ("per step of rank 0: the source rows of its clips that carry taps (2 x the resized height, DESIGN.md 4.4) / the "
                              "summed HIP-event time of its launches" if fw.downscale else
                              "per step of rank 0: 3 B/px of its clips / the summed HIP-event time of its launches")},
        "pipeline": ("the reference's default (SceneManager.auto_downscale: what detect(video, detector) and benchmark/__main__.py run)"
                     if fw.downscale else "full resolution (auto_downscale off)"),
        "cuts_found": sum(len(v) for r in (fw.result or []) for v in r.values()),
        "ground_truth": fw.truth_f1(),
    }
    if args.no_cpu_baseline:
        out["cpu_baseline"] = None
    else:
        out["parity_sample"], out["cpu_baseline"] = fw.parity(args.cpu_sample // 4, os.cpu_count() or 1)
    return out


def respawn_one_rank_per_gpu(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N ranks of THIS script under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1, a free port) and hand their exit code back.  The ranks inherit
    stdout, and only rank 0 prints, so the caller still sees exactly one JSON line."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    env["PSD_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(sys.argv[0]), *argv]
    return subprocess.call(cmd, env=env)


def init_ranks(on_gpu: bool):
    """(world, rank, local_rank, use_dist).  Under a launcher (RANK set) the process group is always created, even for
    one rank, so the exchange path is the same code for N = 1 (launched that way) and N = 8."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    return world, rank, local_rank, use_dist


def main(argv=None, engine_factory=None, cpu_dry_run: bool = False) -> None:
    """``engine_factory``/``cpu_dry_run`` exist for tests/test_bench_plumbing.py only: they run this very
    control flow (pipelining, exchange, JSON) on CPU tensors over gloo with a stand-in engine, so the
    N > 1 path is exercised without GPUs.  The measured path always uses the HIP engine on cuda."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU batch")
    ap.add_argument("--dist", default="U", choices=["U", "K", "S", "T"])
    ap.add_argument("--detector", default="content", choices=["content", "hist", "all", "hash", "edges"],
                    help="content = ContentDetector (headline); hist = Histogram+Threshold; all = all four fused; "
                         "hash = HashDetector (thumbnail kernel + DCT epilogue); edges = ContentDetector with weights (1,1,1,1)")
    ap.add_argument("--res", default="1080p", choices=["1080p", "4k"])
    ap.add_argument("--downscale", default=None, help="'auto' (the reference's default: to about 256 px width) or a factor")
    ap.add_argument("--cpu-sample", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other BASELINE configurations")
    ap.add_argument("--secondary-frames", type=int, default=2048)
    ap.add_argument("--workload", default="headline", choices=["headline", "corpus", "bbc"],
                    help="headline = one resident batch (BASELINE configs[1], or what --detector/--res/--downscale select); "
                         "corpus = configs[4] (mixed 1080p/4K clips, four detectors, sharded by clip); bbc = configs[3]")
    ap.add_argument("--alt-streams", action="store_true", help=argparse.SUPPRESS)   # experiment: consecutive submissions alternate between two streams
    ap.add_argument("--flow-pipeline", default="default", choices=["default", "full"],
                    help="--workload corpus|bbc: default = every clip behind the reference's auto-downscale (what detect() scores); "
                         "full = full-resolution frames (auto_downscale off; the form rounds 2-5 timed)")
    ap.add_argument("--flow-corpus-frames", type=int, default=2048, help="length of the corpus flow inside the default run's `secondary`")
    ap.add_argument("--flow-bbc-frames", type=int, default=6000, help="length of the BBC flow inside the default run's `secondary`")
    ap.add_argument("--corpus-frames", type=int, default=2048, help="frames per 1080p clip of --workload corpus (4K clips: a quarter)")
    ap.add_argument("--bbc-frames", type=int, default=6000, help="frames of the shortest of the 11 clips of --workload bbc")
    ap.add_argument("--exchange", default="default", choices=["default", "step", "stream", "inline", "off"], help=argparse.SUPPRESS)
    # The score-vector all-gather under a launcher.  Clips are independent, so the data path has NO collective: every rank
    # scores and decides its own batch each step (the step ends with that rank's cut list); the ranks' score vectors only have
    # to meet once so that every rank holds the whole job's result.  default = ONE all-gather of every timed step's vectors at
    # the end of the run, inside the timed region (32 bytes per frame, taken from the records the engine has copied to the host
    # anyway: no GPU operation beside the scoring kernels).  step = the round-1..3 form: an all-gather of the device-resident
    # vectors every step (stream / inline: the same on a side stream / in line with the scoring) -- at one rank its three small
    # operations (slice kernel, all-gather, device -> host copy; 22 us of GPU time) stretch the concurrently running HSV
    # kernel by 0.12-0.3 ms = 2.4-5 % whichever stream they run on (profiles/r03_y_*, r04_n_*), which is why it is not the
    # default any more.  off = no exchange at all (diagnosis).
    ap.add_argument("--height", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--width", type=int, default=0, help=argparse.SUPPRESS)
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        # nobody launched the ranks for us: do it ourselves (one process per GPU) and pass their result through
        sys.exit(respawn_one_rank_per_gpu(args.gpus, argv))
    H, W = (2160, 3840) if args.res == "4k" else (1080, 1920)
    if args.height and args.width:
        H, W = args.height, args.width
    on_gpu = not cpu_dry_run

    world, rank, local_rank, use_dist = init_ranks(on_gpu)
    if args.gpus != world and rank == 0:
        print(f"bench: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    device = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(device)

    from pyscenedetect_amd import engine as E
    from pyscenedetect_amd import epilogue

    eng = engine_factory(local_rank) if engine_factory else E.ScoringEngine(local_rank)
    if args.workload != "headline":
        out = run_flow(args, eng, device, world, rank, local_rank, use_dist, on_gpu)
        eng.close()
        if use_dist:
            import torch.distributed as dist

            dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()
            bad = parity_failures(out)
            if bad:
                print("bench: parity sample(s) differ from the oracle: " + "; ".join(bad), file=sys.stderr)
                sys.exit(3)
        return
    n = args.frames
    batch = make_batch(n, args.dist, 20250921 + rank, device, H, W)
    import contextlib

    # --exchange stream: the exchange on a stream of its own instead of torch's current (legacy default) stream.
    # --exchange inline: scoring AND the exchange's small operations on ONE torch stream (the engine takes any stream), so
    # that the slice kernel and the device -> host copy of the gathered vectors run between two scoring kernels instead of
    # beside one (beside one, each of them stretches it by ~40 us: profiles/r03_y_*); only the collective itself still
    # overlaps the scoring.  The gathered vectors then reach the host through a pinned buffer, checked one step later.
    # Measured at one rank (profiles/r03_ah_*): kernel 5.00 ms without the exchange, 5.14 ms with it (default), 5.10 ms inline --
    # not enough to change the default path.
    inline = on_gpu and use_dist and args.exchange == "inline"
    per_step = use_dist and args.exchange in ("step", "stream", "inline")
    deferred = use_dist and args.exchange == "default"
    xstream = torch.cuda.Stream(device) if (on_gpu and use_dist and args.exchange in ("stream", "inline")) else None
    wl = Workload(eng, batch, args.detector, args.downscale, epilogue, E, stream=xstream.cuda_stream if inline else None)
    if args.alt_streams and on_gpu:
        keep_alt = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
        wl.alt_streams = [st.cuda_stream for st in keep_alt]

    kernel_ms: list[float] = []
    state = {"pending_gather": None, "gathered": None, "copies": [], "pinned": [], "held": [], "exchanges": 0}

    def on_xstream():
        return torch.cuda.stream(xstream) if xstream is not None else contextlib.nullcontext()

    def resolve_copies(keep: int):
        """Host side of the inline exchange: gathered vectors whose copy has been issued at least `keep` steps ago."""
        while len(state["copies"]) > keep:
            ev, host, mine = state["copies"].pop(0)
            ev.synchronize()
            allv = host.numpy().astype(np.uint64).reshape(world, -1, 4)
            assert np.array_equal(allv[rank, :, 0], mine)
            state["gathered"] = allv
            state["pinned"].append(host)

    def consume_gather(final: bool = False):
        """Finish the score-vector all-gather issued one step earlier (keeps ranks loosely coupled)."""
        pend = state["pending_gather"]
        if pend is not None:
            work, recv, mine = pend
            with on_xstream():
                work.wait()
                if inline:
                    host = state["pinned"].pop() if state["pinned"] else torch.empty(recv.shape, dtype=recv.dtype, pin_memory=True)
                    host.copy_(recv, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(xstream)
                    state["copies"].append((ev, host, mine))
                else:
                    allv = recv.cpu().numpy().astype(np.uint64).reshape(world, -1, 4)   # every clip's score vectors
                    assert np.array_equal(allv[rank, :, 0], mine)
                    state["gathered"] = allv
            state["pending_gather"] = None
        if inline:
            resolve_copies(0 if final else 1)

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
        if deferred:
            # hold this step's score vectors (host records; 32 bytes per frame) for the run's one all-gather
            recs = wl.state["recs"]
            state["held"].append(np.stack([recs["sad_h"], recs["sad_s"], recs["sad_v"], recs["edge_xor"]], axis=1).astype(np.int64))
        if per_step:
            # score vectors only: 4 x u64 per frame, all-gathered from HBM
            import torch.distributed as dist

            consume_gather()
            recs = wl.state["recs"]
            with on_xstream():
                send = score_vectors_on_device(recs)
                recv = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=device)  # rank-major concat
                work = dist.all_gather_into_tensor(recv, send, async_op=True)
            state["pending_gather"] = (work, recv, recs["sad_h"].copy())

    def run(steps: int, timing: bool):
        if args.detector == "hash":
            wl.run_hash_steps(steps, kernel_ms if timing else [])
            return
        wl.submit()
        for _ in range(steps - 1):
            wl.submit()
            finish(timing)
        finish(timing)
        if per_step:
            consume_gather(final=True)   # the last step's exchange completes inside the timed region
        if deferred:
            exchange_held()              # ... and so does the one all-gather of the deferred form

    def exchange_held():
        """ONE all-gather of the score vectors of every step since the last call: rank-major [world, steps * n, 4]."""
        import torch.distributed as dist

        if not state["held"]:
            return
        mine = np.concatenate(state["held"], axis=0)
        state["held"].clear()
        send = torch.from_numpy(mine).to(device)
        recv = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=device)
        dist.all_gather_into_tensor(recv, send)
        allv = recv.cpu().numpy().astype(np.uint64).reshape(world, -1, 4)
        assert np.array_equal(allv[rank].astype(np.int64), mine)
        state["gathered"] = allv
        state["exchanges"] += 1

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
    # What this box does under this workload, OUTSIDE the timed region: the same steps back to back for about a second and a half
    # while a thread samples socket power and shader clock (the sensors are moving averages: the 0.1 s of the timed steps still
    # show the idle value).  Rank 0 only, and not in the CPU dry run.
    smi = SmiPoll(local_rank if on_gpu else 0)
    if on_gpu and rank == 0 and args.detector != "hash" and elapsed > 0:
        probe_steps = int(min(2000, max(args.steps, 1.5 / (elapsed / args.steps))))
        with smi:
            wl.submit()                       # (the workload's own steps, two in flight, without the launcher's exchange)
            for _ in range(probe_steps - 1):
                wl.submit()
                wl.finish()
            wl.finish()
            torch.cuda.synchronize(device)
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
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                key = f"{args.detector}_{args.res}_{args.dist}_{n}" + (f"_downscale_{args.downscale}" if args.downscale else "")
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
                traffic_source = ("profiles/hbm_traffic.json[%r]: %s -- a committed PMC measurement of this kernel and "
                                  "shape, not counted in this run" % (key, tj[key].get("source", "?"))) if key in tj else None
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
                "frames_per_gpu": n, "height": H, "width": W, "distribution": args.dist, "ranks_seen": world,
                "parallelism": f"clips sharded over {world} GPU(s), no data-path collective; RCCL all-gather of the score vectors"
                               if use_dist else "1 GPU",
                "pipeline_depth": 1 if args.detector == "hash" else 2,
                "exchange": None if not use_dist else
                            ("one all-gather of all %d timed steps' score vectors (32 B/frame) at the end of the timed region; "
                             "no collective on the data path (clips are independent)" % args.steps) if deferred else
                            "off" if args.exchange == "off" else
                            ("every step: host copy of the records (" + state["no_device_view"] + ")") if state.get("no_device_view") else
                            "every step: score vectors sliced out of the device-resident records (psd_last_records_device)",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": wl.kernel_name(),
                "avg_launch_ms": round(avg_kernel_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "limiter": "package power: the HSV pass holds the 1400 W cap and the shader clock falls to 1.9-2.1 GHz "
                           "(profiles/r02_a_power_and_clock_by_build.txt)" if headline else None,
                # what THIS box does under this workload (power-bound passes read 4 - 6 % apart between boxes of the pool): sampled
                # during ~1.5 s of the same steps run again behind the timed region
                "box": smi.summary(),
            },
            "cuts_found": len(wl.state["cuts"]),
        }
        if not args.no_cpu_baseline and world == 1 and args.detector == "hash":
            from oracle import lib as orc

            sample = batch[: min(64, n)].cpu().numpy()
            t0 = time.perf_counter()
            ref = orc.hash_bits(orc.hash_thumbs(sample, wl.HASH_SIZE * wl.HASH_LOWPASS), wl.HASH_SIZE)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": round(len(sample) / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same batch through oracle/cv2_restate.c (grey + INTER_AREA + DCT + median), "
                                             "one thread; real OpenCV is not installed" % len(sample)}
            out["parity_sample"] = parity_sample(wl, min(16, n))
        elif not args.no_cpu_baseline and world == 1 and not args.downscale and args.detector != "edges":
            sample = batch[: args.cpu_sample].cpu().numpy()
            cb = cpu_baseline(sample, wl.flags & 7, os.cpu_count() or 1)
            ref = cb.pop("_records")
            got = wl.state["recs"]
            fields = [f for f in ("sad_h", "sad_s", "sad_v", "byte_sum", "hist") if f in got.dtype.names]
            same = all(np.array_equal(got[f][: args.cpu_sample], ref[f]) for f in fields)
            # ... and deep into the batch: the middle and the last 64 frames, both sides of every boundary between two time
            # walks of the launch, both sides of the 2^32 / 2^33 / 2^34 byte offsets (frame pointers beyond 32 bits)
            runs = [r for r in parity_runs(n, 64, wl.walk, H * W * 3) if r[1] > args.cpu_sample]
            wants = oracle_records_at(batch, runs, wl.flags, os.cpu_count() or 1)
            same = same and all(np.array_equal(got[f][a:b], want[f]) for (a, b), want in zip(runs, wants) for f in fields)
            out["cpu_baseline"] = cb
            what = "records of the first %d frames and of %s" % (args.cpu_sample, describe_runs(runs, wl.walk))
            out["parity_sample"] = (what + " identical to the oracle") if same else "MISMATCH vs oracle: " + what
        else:
            out["cpu_baseline"] = None
        if on_gpu and world == 1 and not use_dist and not args.no_secondary and headline:
            out["secondary"] = secondary_runs(eng, batch, E, epilogue, device, min(args.secondary_frames, n), args)
    eng.close()
    if use_dist:
        import torch.distributed as dist

        dist.barrier(device_ids=[local_rank]) if on_gpu else dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
        bad = parity_failures(out)
        if bad:
            print("bench: parity sample(s) differ from the oracle: " + "; ".join(bad), file=sys.stderr)
            sys.exit(3)


def parity_failures(obj, path: str = "") -> list[str]:
    """Every `parity_sample` anywhere in the result that reports a MISMATCH (the line is printed first, then bench.py exits
    non-zero: a fast number whose records differ from the reference's is not a result)."""
    bad = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            if k == "parity_sample" and isinstance(v, str) and "MISMATCH" in v:
                bad.append((path or "headline") + ": " + v[:160])
            else:
                bad += parity_failures(v, f"{path}.{k}" if path else k)
    return bad


if __name__ == "__main__":
    main()
