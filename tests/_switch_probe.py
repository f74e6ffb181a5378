"""Run by tests/test_gpu_switches.py in a subprocess (the library reads its PSD_* environment switches once per process):
every entry point a switch can reroute, on small inputs, against the oracle.  Prints "ok" or dies on the first mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.append(os.path.join(ROOT, "oracle", "cv2_shim"))

import cv2  # noqa: E402  (the oracle's shim)

from oracle import lib as orc  # noqa: E402
from oracle.detectors_np import score_batch as oracle_score  # noqa: E402
from pyscenedetect_amd import engine as E  # noqa: E402
from pyscenedetect_amd.synth import make_clip  # noqa: E402

FIELDS = ("sad_h", "sad_s", "sad_v", "byte_sum", "edge_xor")


def same(got, want, fields, tag):
    for f in fields:
        assert np.array_equal(got[f], want[f]), (tag, f, np.flatnonzero(np.asarray(got[f]) != np.asarray(want[f]))[:8])


def main():
    eng = E.ScoringEngine(0)
    # 256 x 144 (16-byte aligned frames: the staged kernels), long enough for several time chunks and flushes
    frames, _ = make_clip(77, 150, 144, 256, shot_len=(20, 40))
    want = oracle_score(frames, edges=True)
    got = eng.score_host(frames, flags=E.SCORE_ALL)
    same(got, want, FIELDS, "all")
    assert np.array_equal(got["hist"], want["hist"]), "all: hist"
    for flags, fields, hist in ((E.SCORE_HSV_SAD, FIELDS[:3], False), (E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM, ("byte_sum",), True),
                                (E.SCORE_HSV_SAD | E.SCORE_EDGES, FIELDS[:3] + ("edge_xor",), False), (E.SCORE_EDGES, ("edge_xor",), False),
                                (E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM, FIELDS[:4], True)):
        got = eng.score_host(frames[1:], prev=frames[0], flags=flags)
        same(got, want[1:], fields, flags)
        if hist:
            assert np.array_equal(got["hist"], want["hist"][1:]), (flags, "hist")
    # a weak-edge chain across dozens of 64x64 tiles: more hysteresis launches than the speculative path enqueues
    from tests.test_gpu_fullsize import serpentine

    pair = np.stack([serpentine(300, 700, seed=False)[0], serpentine(300, 700)[0]])
    ws = oracle_score(pair, edges=True, kernel_size=3)
    assert ws["edge_xor"][1] > 1000, "the chain must be there"
    same(eng.score_host(pair, flags=E.SCORE_EDGES, edge_kernel=3), ws, ("edge_xor",), "serpentine")
    same(eng.score_host(pair, flags=E.SCORE_EDGES | E.SCORE_HSV_SAD, edge_kernel=3), ws, FIELDS[:3] + ("edge_xor",), "serpentine + hsv")
    # HashDetector thumbnails
    big, _ = make_clip(5, 6, 360, 640, shot_len=(2, 3))
    assert np.array_equal(eng.hash_thumbs_host(big, 16), orc.hash_thumbs(big, 16)), "hash thumbnails"
    assert np.array_equal(eng.hash_thumbs_host(frames[:5], 32), orc.hash_thumbs(frames[:5], 32)), "hash thumbnails (144 x 256)"
    # the default pipeline: downscale fused with the HSV term, and downscale into a buffer + the other terms
    buf = eng.alloc(big.nbytes)
    buf.upload(big.reshape(-1))
    small = np.stack([cv2.resize(f, (256, 144)) for f in big])
    w2 = orc.score_batch(small)
    same(eng.score_device_downscaled(buf.ptr, len(big), 360, 640, 144, 256, flags=E.SCORE_HSV_SAD), w2, FIELDS[:3], "downscaled hsv")
    got = eng.score_device_downscaled(buf.ptr, len(big), 360, 640, 144, 256, flags=E.SCORE_ALL & ~E.SCORE_EDGES)
    same(got, w2, FIELDS[:4], "downscaled all")
    assert np.array_equal(got["hist"], w2["hist"]), "downscaled hist"
    # ... with the edge term: the resized frames go through the engine's buffer (the storing instance of the downscale kernel)
    w3 = oracle_score(small, edges=True)
    same(eng.score_device_downscaled(buf.ptr, len(big), 360, 640, 144, 256, flags=E.SCORE_HSV_SAD | E.SCORE_EDGES), w3, FIELDS[:3] + ("edge_xor",),
         "downscaled hsv + edges")
    out = eng.alloc(small.nbytes)
    eng.resize_device(buf.ptr, len(big), 360, 640, out.ptr, 144, 256)
    eng.synchronize()
    assert np.array_equal(out.download(small.nbytes).reshape(small.shape), small), "resize_device"
    out.free()
    buf.free()
    # INTER_CUBIC in the form PSD_CUBIC_FORM names (the shim reads the same switch): noise that tells the three forms apart, a destination
    # width whose rows end in a scalar tail
    noise = np.random.default_rng(11).integers(0, 256, (8, 360, 640, 3), dtype=np.uint8)
    form = cv2.CUBIC_FORMS[os.environ.get("PSD_CUBIC_FORM", "sse")]
    nbuf = eng.alloc(noise.nbytes)
    nbuf.upload(noise.reshape(-1))
    told_apart = set()
    for dh, dw in ((144, 256), (143, 255)):
        forms = []
        for f in range(3):
            o = np.empty((len(noise), dh, dw, 3), np.uint8)
            for i in range(len(noise)):
                orc.lib().orc_resize_cubic_u8(noise[i].ctypes.data, 640 * 3, 360, 640, 3, o[i].ctypes.data, dw * 3, dh, dw, f)
            forms.append(o)
        told_apart |= {(a, b) for a, b in ((0, 1), (0, 2), (1, 2)) if not np.array_equal(forms[a], forms[b])}
        out = eng.alloc(forms[0].nbytes)
        eng.resize_device(nbuf.ptr, len(noise), 360, 640, out.ptr, dh, dw, interpolation=cv2.INTER_CUBIC)
        eng.synchronize()
        got = out.download(forms[0].nbytes).reshape(forms[0].shape)
        assert np.array_equal(got, forms[form]), ("INTER_CUBIC", form, dh, dw, int(np.count_nonzero(got != forms[form])))
        assert np.array_equal(got[0], cv2.resize(noise[0], (dw, dh), interpolation=cv2.INTER_CUBIC)), "the shim follows the same switch"
        out.free()
    assert len(told_apart) == 3, ("the input must tell the three forms apart", told_apart)
    nbuf.free()
    # submissions with LARGE record sets, two in flight (the packed-heads form of >= 256 records, the whole-record copy): 320
    # records with histograms (340 KB), 7000 without (280 KB of heads)
    many = np.random.default_rng(3).integers(0, 256, (7000, 16, 32, 3), dtype=np.uint8)
    many[3000:] //= 3
    wm = orc.score_batch(many)
    dbuf = eng.alloc(many.nbytes)
    dbuf.upload(many.reshape(-1))
    for n, flags, sums in ((320, E.SCORE_ALL & ~E.SCORE_EDGES, False), (7000, E.SCORE_HSV_SAD, True)):
        eng.submit_device(dbuf.ptr, n, 16, 32, flags=flags)
        eng.submit_device(dbuf.ptr, n, 16, 32, flags=flags)
        for rep in range(2):
            got = eng.collect(n, sums_only=sums)
            same(got, wm[:n], FIELDS[:3] + (() if sums else ("byte_sum",)), ("large records", n, rep))
            if not sums:
                assert np.array_equal(got["hist"], wm["hist"][:n]), ("large records: hist", rep)
    dbuf.free()
    eng.close()
    print("ok")


if __name__ == "__main__":
    main()
