"""Differential fuzz of the HIP engine against the CPU oracle (runs on the GPU box; the oracle is the checker).

    python tools/fuzz_gpu.py [--seconds 60] [--seed 1] [--max-cases 100000]

Every case draws a frame shape (1 x 1 ... about 300 x 300, biased towards the edges of the kernels' fast paths: widths
around multiples of 4 / 16, pixel counts around multiples of 16, single rows and columns), a batch length, a content
kind (noise, smooth shots, flat, two-level, grey, saturated primaries -- the hue's tie cases), a memory layout
(contiguous, padded rows, padded frames, an offset base), a set of terms, a predecessor or none, and one of the entry
points (score_host, score_frames, score_host behind a downscale with each interpolation mode, score_clips with packed
clips, hash thumbnails), and requires records identical to `oracle.detectors_np.OracleEngine` / `oracle.lib`.
Prints one JSON line: cases run per entry point, the first mismatches (with the seed and case number that reproduce
them).  tests/test_gpu_fuzz.py runs a short fixed-seed slice of the same generator.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.append(os.path.join(ROOT, "oracle", "cv2_shim"))

import numpy as np  # noqa: E402

FIELDS = ("sad_h", "sad_s", "sad_v", "edge_xor", "byte_sum", "hist")


BIG = False      # --big: video-sized frames (720p ... 8K, also a few pixels off those sizes), one to three frames per case


def draw_shape(rng) -> tuple[int, int]:
    if BIG:
        h, w = [(720, 1280), (1080, 1920), (1080, 1920), (2160, 3840), (1440, 2560), (576, 720), (4320, 7680), (1088, 1920)][int(rng.integers(0, 8))]
        if (h, w) == (4320, 7680) and rng.integers(0, 3):
            h, w = 1080, 1920
        if rng.integers(0, 3) == 0:
            h, w = h + int(rng.integers(-3, 4)), w + int(rng.integers(-5, 6))
        return h, w
    kind = rng.integers(0, 8)
    if kind == 0:      # tiny
        return int(rng.integers(1, 9)), int(rng.integers(1, 9))
    if kind == 1:      # one row / one column
        return (1, int(rng.integers(1, 400))) if rng.integers(0, 2) else (int(rng.integers(1, 400)), 1)
    if kind == 2:      # pixel count around a multiple of 16
        w = int(rng.integers(3, 200))
        h = int(rng.integers(1, 64))
        return h, w
    if kind == 3:      # width around a multiple of 4 / 16 / 32 / 64 / 128
        base = int(rng.choice([4, 16, 32, 64, 128, 256]))
        w = max(1, base * int(rng.integers(1, 4)) + int(rng.integers(-2, 3)))
        return int(rng.integers(2, 150)), w
    if kind == 4:      # tall and narrow
        return int(rng.integers(100, 500)), int(rng.integers(1, 12))
    if kind == 5:      # around the hysteresis / Sobel tile sizes (64 x 64, 128 x 32)
        return int(rng.choice([31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 129])), int(rng.choice([63, 64, 65, 127, 128, 129, 191, 192, 193, 256, 257]))
    return int(rng.integers(2, 300)), int(rng.integers(2, 300))


def draw_content(rng, n: int, h: int, w: int) -> np.ndarray:
    kind = rng.integers(0, 8)
    if kind == 0:
        return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if kind == 1:      # flat frames, a different colour each
        return np.broadcast_to(rng.integers(0, 256, (n, 1, 1, 3), dtype=np.uint8), (n, h, w, 3)).copy()
    if kind == 2:      # grey: diff == 0 everywhere (hdiv[0], sdiv[v])
        g = rng.integers(0, 256, (n, h, w, 1), dtype=np.uint8)
        return np.repeat(g, 3, axis=3)
    if kind == 3:      # two-level images with blocks: many Canny edges, chains across tiles
        bs = int(rng.integers(1, 24))
        yy, xx = np.mgrid[0:h, 0:w]
        out = np.empty((n, h, w, 3), np.uint8)
        for t in range(n):
            m = (((yy + t * int(rng.integers(0, 3))) // bs + (xx + t) // bs) % 2).astype(np.uint8)
            lo, hi = rng.integers(0, 100, 3), rng.integers(120, 256, 3)
            out[t] = np.where(m[..., None] > 0, hi, lo).astype(np.uint8)
        return out
    if kind == 4:      # saturated primaries and their ties (v == r == g, ...)
        pal = np.array([[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 0], [0, 255, 255], [255, 0, 255], [255, 255, 255], [0, 0, 0],
                        [128, 128, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [254, 255, 255], [255, 254, 255]], np.uint8)
        return pal[rng.integers(0, len(pal), (n, h, w))]
    if kind == 5:      # smooth shots + noise + a cut in the middle
        gy, gx = np.linspace(0, 1, h)[:, None, None], np.linspace(0, 1, w)[None, :, None]
        out = np.empty((n, h, w, 3), np.uint8)
        a, b = rng.uniform(0, 255, (2, 3)), rng.uniform(0, 255, (2, 3))
        for t in range(n):
            c = a if t < n // 2 else b
            img = c[0] * gy + c[1] * gx * (1 - gy) + rng.normal(0, 2, (h, w, 3))
            out[t] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        return out
    if kind == 6:      # dark frames: median 0 / 1 (Canny low threshold 0)
        return rng.integers(0, 3, (n, h, w, 3), dtype=np.uint8)
    # a bright line drifting over noise (weak chains that cross tiles)
    out = rng.integers(90, 110, (n, h, w, 3), dtype=np.uint8)
    for t in range(n):
        y = (np.arange(w) * max(1, h) // max(1, w) + 3 * t) % h
        out[t, y, np.arange(w)] = rng.integers(130, 256)
    return out


def relayout(rng, frames: np.ndarray) -> np.ndarray:
    """The same frames in a different memory layout (a view; values unchanged)."""
    n, h, w, _ = frames.shape
    kind = rng.integers(0, 5)
    if kind == 0 or n == 0:
        return frames
    if kind == 1:      # padded rows
        pad = int(rng.integers(1, 40))
        big = np.zeros((n, h, w * 3 + pad), np.uint8)
        big[:, :, : w * 3] = frames.reshape(n, h, w * 3)
        return np.lib.stride_tricks.as_strided(big, (n, h, w, 3), (big.strides[0], big.strides[1], 3, 1))
    if kind == 2:      # padded frames
        pad = int(rng.integers(1, 100))
        big = np.zeros((n, h * w * 3 + pad), np.uint8)
        big[:, : h * w * 3] = frames.reshape(n, -1)
        return np.lib.stride_tricks.as_strided(big, (n, h, w, 3), (big.strides[0], w * 3, 3, 1))
    if kind == 3:      # misaligned base
        off = int(rng.integers(1, 16))
        big = np.zeros(frames.size + 16, np.uint8)
        big[off:off + frames.size] = frames.reshape(-1)
        return big[off:off + frames.size].reshape(frames.shape)
    # every second frame of a longer array
    big = np.zeros((2 * n, h, w, 3), np.uint8)
    big[::2] = frames
    return big[::2]


def compare(got, want, fields) -> list[str]:
    return [f for f in fields if not np.array_equal(got[f], want[f])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--big", action="store_true", help="video-sized frames, up to three per case")
    args = ap.parse_args()
    global BIG
    BIG = args.big
    out = run(args.seed, args.seconds, args.max_cases, args.verbose)
    print(json.dumps(out))
    return 1 if out["mismatches"] else 0


def run(seed: int, seconds: float, max_cases: int, verbose: bool = False, engine=None) -> dict:
    from oracle import lib as orc
    from oracle.detectors_np import OracleEngine
    from pyscenedetect_amd import engine as E

    eng = engine or E.ScoringEngine(0)
    ora = OracleEngine()
    counts: dict[str, int] = {}
    bad: list[dict] = []
    t_end = time.perf_counter() + seconds
    case = 0
    while case < max_cases and time.perf_counter() < t_end and len(bad) < 20:
        rng = np.random.default_rng([seed, case])
        h, w = draw_shape(rng)
        entry = str(rng.choice(["score_host", "score_host", "score_frames", "downscale", "clips", "hash", "edges", "edges"]))
        n = int(rng.integers(1, 9)) if rng.integers(0, 4) else int(rng.integers(9, 40))
        if h * w > 40000:
            n = min(n, 6)
        if BIG:
            n = int(rng.integers(1, 4)) if h * w < 5_000_000 else 1
        desc = {"seed": seed, "case": case, "entry": entry, "h": h, "w": w, "n": n}
        try:
            if entry in ("score_host", "score_frames", "edges"):
                flags = int(rng.integers(1, 8))
                kernel = 0
                if entry == "edges":
                    flags |= 8 if rng.integers(0, 4) else 0
                    flags = (flags & 9) or 8 if rng.integers(0, 2) else flags | 8
                    kernel = int(rng.choice([0, 0, 3, 5, 7, 13, 21, 63]))
                    n = min(n, 6)
                    if BIG:
                        n = min(n, 2 if h * w < 3_000_000 else 1)
                frames = draw_content(rng, n, h, w)
                prev = draw_content(rng, 1, h, w)[0] if rng.integers(0, 2) else None
                desc.update(flags=flags, kernel=kernel, prev=prev is not None)
                view = relayout(rng, frames)
                if entry == "score_frames":
                    got = eng.score_frames([view[i] for i in range(n)], prev, flags=flags, edge_kernel=kernel)
                else:
                    got = eng.score_host(view, prev, flags=flags, edge_kernel=kernel)
                want = ora.score_host(frames, prev, flags=flags, edge_kernel=kernel)
                fields = [f for f in FIELDS if (f in ("sad_h", "sad_s", "sad_v") and flags & 1) or (f in ("hist", "byte_sum") and flags & 6)
                          or (f == "edge_xor" and flags & 8)]
                diff = compare(got, want, fields)
            elif entry == "downscale":
                factor = float(rng.choice([1.25, 1.5, 2.0, 2.5, 3.0, 4.0, 7.5, rng.uniform(1.05, 9.0)]))
                interp = int(rng.choice([1, 1, 1, 0, 3, 4, 2]))       # INTER_LINEAR, NEAREST, AREA, LANCZOS4, CUBIC (the default form)
                flags = int(rng.integers(1, 8))
                frames = draw_content(rng, n, h, w)
                prev = draw_content(rng, 1, h, w)[0] if rng.integers(0, 2) else None
                desc.update(flags=flags, factor=factor, interp=interp, prev=prev is not None)
                want = ora.score_host(frames, prev, flags=flags, downscale=factor, interpolation=interp)
                if rng.integers(0, 2):
                    got = eng.score_host(frames, prev, flags=flags, downscale=factor, interpolation=interp)
                else:
                    desc["entry"] = "downscale/score_frames"
                    got = eng.score_frames([frames[i] for i in range(n)], prev, flags=flags, downscale=factor, interpolation=interp)
                fields = [f for f in FIELDS if (f in ("sad_h", "sad_s", "sad_v") and flags & 1) or (f in ("hist", "byte_sum") and flags & 6)]
                diff = compare(got, want, fields)
            elif entry == "clips":
                # several clips of one or two sizes, resident, packed by the engine into shared batches
                flags = int(rng.integers(1, 8))
                sizes = [(h, w)] + ([draw_shape(rng)] if rng.integers(0, 2) else [])
                # round 6: half of the cases behind the resize the reference's SceneManager puts in front of its detectors
                # (psd_score_segments_downscaled_device): its default "auto" or a factor, each interpolation, now and then the edge term
                ds, interp = None, 1
                if rng.integers(0, 2):
                    ds = "auto" if rng.integers(0, 2) else float(rng.choice([1.25, 1.5, 2.0, 2.5, 3.0, 7.5, rng.uniform(1.05, 6.0)]))
                    interp = int(rng.choice([1, 1, 1, 0, 3, 4, 2]))
                    if rng.integers(0, 5) == 0:
                        flags |= 8
                clips, wants = [], []
                for _ in range(int(rng.integers(1, 6)) if not BIG else 2):
                    ch, cw = sizes[int(rng.integers(0, len(sizes)))]
                    cn = int(rng.integers(1, 12)) if not BIG else int(rng.integers(1, 3))
                    fr = draw_content(rng, cn, ch, cw)
                    clips.append(fr)
                    wants.append(ora.score_host(fr, None, flags=flags, downscale=E.downscale_size(ch, cw, ds)[0], interpolation=interp))
                desc.update(flags=flags, clips=[c.shape[:3] for c in clips], downscale=ds, interp=interp)
                if ds is not None:
                    desc["entry"] = "clips/downscaled" 
                import torch

                if torch.cuda.is_available():
                    # a mix: host arrays, separately allocated device tensors, and device clips sitting back to back
                    # in one allocation (scored in place as one run)
                    dev = []
                    packed = [i for i, c in enumerate(clips) if c.shape[1:3] == clips[0].shape[1:3]] if rng.integers(0, 2) else []
                    pool = torch.from_numpy(np.concatenate([clips[i] for i in packed])).cuda() if packed else None
                    off = 0
                    for i, c in enumerate(clips):
                        if i in packed:
                            dev.append(pool[off:off + c.shape[0]])
                            off += c.shape[0]
                        elif rng.integers(0, 2):
                            dev.append(torch.from_numpy(c).cuda())
                        else:
                            dev.append(c)
                else:
                    dev = clips
                gots = eng.score_clips(dev, flags=flags, downscale=ds, interpolation=interp)
                fields = [f for f in FIELDS if (f in ("sad_h", "sad_s", "sad_v") and flags & 1) or (f in ("hist", "byte_sum") and flags & 6)
                          or (f == "edge_xor" and flags & 8)]
                diff = []
                for g, wnt in zip(gots, wants):
                    diff += [f for f in fields if f in g.dtype.names and not np.array_equal(g[f], wnt[f])]
                del dev
            else:  # hash
                size = int(rng.choice([8, 16, 16, 32]))
                if (h < size or w < size) and rng.integers(0, 2):      # (the other half stays: thumbnails LARGER than the frame along
                    h, w = h + size, w + size                             #  an axis -- OpenCV's enlarging INTER_AREA, round 5)
                    desc.update(h=h, w=w)
                frames = draw_content(rng, n, h, w)
                desc.update(size=size)
                got = eng.hash_thumbs_host(frames, size)
                want = orc.hash_thumbs(frames, size)
                diff = [] if np.array_equal(got, want) else ["thumbs"]
        except NotImplementedError as ex:      # a shape the engine refuses by contract (stated in the message)
            counts["refused:" + desc["entry"]] = counts.get("refused:" + desc["entry"], 0) + 1
            if verbose:
                print("refused", desc, ex, file=sys.stderr)
            case += 1
            continue
        except Exception as ex:  # noqa: BLE001
            diff = ["%s: %s" % (type(ex).__name__, str(ex)[:200])]
        counts[desc["entry"]] = counts.get(desc["entry"], 0) + 1
        if diff:
            desc["differs"] = diff
            desc = {k: (v if not isinstance(v, (np.integer, np.floating)) else v.item()) for k, v in desc.items()}
            bad.append(json.loads(json.dumps(desc, default=str)))
            if verbose:
                print("MISMATCH", desc, file=sys.stderr)
        case += 1
    return {"seed": seed, "cases": case, "by_entry": counts, "mismatches": bad}


if __name__ == "__main__":
    sys.exit(main())
