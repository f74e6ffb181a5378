"""How many HashDetector bits depend on the precision / order of the DCT?  (round-5 review, weak 8)

The reference hashes a frame with ``cv2.dct`` on a float32 image (hash_detector.py:139): OpenCV evaluates that through its DFT code in
float32, in an order that depends on its build and that no restatement can pin.  This engine (``psd_epilogue_hash_bits``) and the oracle
evaluate the orthonormal DCT-II in float64 and round once to float32.  A hash bit is ``dct_low[u, v] > median``; it can only differ between
two correct evaluations where a coefficient lies within rounding noise of the median.  This tool counts, over a corpus of thumbnails, the
bits that differ between the float64 evaluation and three float32 ones:

  seq32   the same matrix products accumulated sequentially in float32
  pair32  float32 with numpy's pairwise / blocked summation (``numpy.matmul`` on float32)
  fft32   SciPy's pocketfft in single precision (``scipy.fft.dct(x, type=2, norm="ortho")`` along both axes): an FFT-based float32 DCT
          by another author -- the closest available stand-in for "some other correct float32 implementation" such as OpenCV's

and what that does to the DECISIONS (Hamming distances across the 0.35 threshold).  CPU only; thumbnails from the oracle.

    python tools/hash_dct_f32_vs_f64.py [--frames 4000] > profiles/r06_hash_dct_f32_vs_f64.txt
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.fft  # noqa: E402

from oracle import lib as orc  # noqa: E402
from pyscenedetect_amd.synth import make_clip, make_clip_fast  # noqa: E402


def basis(size: int, keep: int, dtype) -> np.ndarray:
    k = np.arange(keep)[:, None].astype(np.float64)
    j = np.arange(size)[None, :].astype(np.float64)
    c = np.sqrt(2.0 / size) * np.cos(np.pi * (2 * j + 1) * k / (2.0 * size))
    c[0] = np.sqrt(1.0 / size)
    return c.astype(dtype)


def low_f64(x: np.ndarray, keep: int) -> np.ndarray:
    c = basis(x.shape[0], keep, np.float64)
    return (c @ x.astype(np.float64) @ c.T).astype(np.float32)


def low_seq32(x: np.ndarray, keep: int) -> np.ndarray:
    size = x.shape[0]
    c = basis(size, keep, np.float32)
    tmp = np.zeros((keep, size), np.float32)
    for y in range(size):                                  # sequential float32 accumulation over y, then over x
        tmp += c[:, y:y + 1] * x[y:y + 1, :]
    out = np.zeros((keep, keep), np.float32)
    for xx in range(size):
        out += tmp[:, xx:xx + 1] * c[:, xx][None, :]
    return out


def low_pair32(x: np.ndarray, keep: int) -> np.ndarray:
    c = basis(x.shape[0], keep, np.float32)
    return (c @ x.astype(np.float32) @ c.T).astype(np.float32)


def low_fft32(x: np.ndarray, keep: int) -> np.ndarray:
    full = scipy.fft.dct(scipy.fft.dct(x.astype(np.float32), type=2, norm="ortho", axis=0), type=2, norm="ortho", axis=1)
    assert full.dtype == np.float32
    return full[:keep, :keep]


def bits_of(low: np.ndarray) -> np.ndarray:
    return low > np.median(np.asarray(low, dtype=np.float32))


def corpus(n_target: int):
    """Thumbnail sources: shot-like clips with fades and noise, clips with flat objects, uniform noise, near-constant frames."""
    rng = np.random.default_rng(20250930)
    out, n = [], 0
    seed = 100
    while n < n_target:
        kind = seed % 4
        if kind == 0:
            frames, _ = make_clip(seed, 120, 90, 160, shot_len=(10, 40))
        elif kind == 1:
            frames, _ = make_clip_fast(seed, 160, 180, 320, shot_len=(12, 50))
        elif kind == 2:
            frames = rng.integers(0, 256, (60, 72, 128, 3), dtype=np.uint8)
        else:
            base = rng.integers(0, 256, 3)
            frames = np.clip(base[None, None, None, :] + rng.integers(-2, 3, (80, 72, 128, 3)), 0, 255).astype(np.uint8)
        out.append(frames)
        n += len(frames)
        seed += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4000)
    a = ap.parse_args()
    rows = []
    for size, lowpass in ((16, 2), (8, 2), (16, 4)):          # HashDetector(size, lowpass): the default, a smaller hash, a wider DCT
        im = size * lowpass
        stats = {k: {"bits": 0, "frames": 0, "max_per_frame": 0, "decisions": 0} for k in ("seq32", "pair32", "fft32")}
        total_bits = total_frames = total_pairs = 0
        margins = []
        for frames in corpus(a.frames):
            thumbs = orc.hash_thumbs(frames, im)
            hashes = {k: [] for k in ("f64", "seq32", "pair32", "fft32")}
            for th in thumbs:
                mx = max(int(th.max()), 1)
                x = (th.astype(np.float32) / np.float32(mx)).astype(np.float32)
                lows = {"f64": low_f64(x, size), "seq32": low_seq32(x, size), "pair32": low_pair32(x, size), "fft32": low_fft32(x, size)}
                ref = bits_of(lows["f64"])
                med = np.median(lows["f64"])
                margins.append(float(np.min(np.abs(lows["f64"] - med)[np.abs(lows["f64"] - med) > 0], initial=np.inf)))
                hashes["f64"].append(ref)
                for k in stats:
                    b = bits_of(lows[k])
                    d = int(np.count_nonzero(b != ref))
                    stats[k]["bits"] += d
                    stats[k]["frames"] += d > 0
                    stats[k]["max_per_frame"] = max(stats[k]["max_per_frame"], d)
                    hashes[k].append(b)
                total_bits += size * size
                total_frames += 1
            # decisions: normalised Hamming distance of consecutive hashes against the default threshold 0.35
            def dist(hs):
                return np.array([np.count_nonzero(hs[i] != hs[i - 1]) / float(size * size) for i in range(1, len(hs))])
            d_ref = dist(hashes["f64"])
            total_pairs += len(d_ref)
            for k in stats:
                stats[k]["decisions"] += int(np.count_nonzero((dist(hashes[k]) >= 0.35) != (d_ref >= 0.35)))
        rows.append({"size": size, "lowpass": lowpass, "dct": f"{im}x{im}", "frames": total_frames, "hash_bits": total_bits, "frame_pairs": total_pairs,
                     "smallest_nonzero_margin_to_median_f64": min(margins),
                     **{k: {"bits_differing": v["bits"], "per_million_bits": round(v["bits"] * 1e6 / total_bits, 2), "frames_affected": v["frames"],
                            "max_bits_in_one_frame": v["max_per_frame"], "threshold_decisions_changed": v["decisions"]} for k, v in stats.items()}})
    print(json.dumps({"what": __doc__.split("\n")[0], "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
