"""Freeze golden vectors from a REAL OpenCV build into tests/golden/cv2_vectors.npz.

Run on any machine that has ``cv2`` (the build and GPU images do not): it records the outputs of the
cv2 primitives on the hot path for seeded inputs, so that machines without OpenCV can pin the oracle
to real OpenCV through a committed fixture (tests/test_oracle.py would then load it).

    python tools/dump_cv2_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import cv2

    if "oracle-shim" in getattr(cv2, "__version__", ""):
        sys.exit("this is the oracle's cv2 shim, not OpenCV: nothing to dump")
    out = {"cv2_version": np.array(cv2.__version__)}
    for seed in range(4):
        img = np.random.default_rng(seed).integers(0, 256, (97, 131, 3), dtype=np.uint8)
        hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
        yuv = cv2.cvtColor(img, cv2.COLOR_BGR2YUV)
        lum = np.ascontiguousarray(hsv[..., 2])
        med = np.median(lum)
        low, high = int(max(0, (1 - 1 / 3) * med)), int(min(255, (1 + 1 / 3) * med))
        canny = cv2.Canny(lum, low, high)
        out[f"hsv{seed}"] = hsv
        out[f"yuv{seed}"] = yuv
        out[f"canny{seed}"] = canny
        out[f"dilate5_{seed}"] = cv2.dilate(canny, np.ones((5, 5), np.uint8))
        h = cv2.calcHist([np.ascontiguousarray(yuv[..., 0])], [0], None, [128], [0, 256])
        out[f"hist128_{seed}"] = h
        out[f"hist128n_{seed}"] = cv2.normalize(h, h).flatten()
        out[f"resize_{seed}"] = cv2.resize(img, (64, 48), interpolation=cv2.INTER_LINEAR)
        out[f"resize_nearest_{seed}"] = cv2.resize(img, (64, 48), interpolation=cv2.INTER_NEAREST)
        out[f"resize_area_{seed}"] = cv2.resize(img, (64, 48), interpolation=cv2.INTER_AREA)
        out[f"resize_lanczos4_{seed}"] = cv2.resize(img, (64, 48), interpolation=cv2.INTER_LANCZOS4)
        out[f"resize_cubic_{seed}"] = cv2.resize(img, (61, 47), interpolation=cv2.INTER_CUBIC)      # (183 elements per row: a scalar tail of 7; which
        #  of the three forms of oracle/cv2_restate.c: orc_resize_cubic_u8 this build computes -- if any: IPP builds compute none -- tests/test_real_cv2.py says)
        # HashDetector.hash_frame (hash_detector.py:117-151)
        gray = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
        out[f"gray{seed}"] = gray
        for size in (16, 32):
            small = cv2.resize(gray, (size, size), interpolation=cv2.INTER_AREA)
            out[f"gray_area{size}_{seed}"] = small
            out[f"dct{size}_{seed}"] = cv2.dct(np.float32(small) / max(1, int(small.max())))
    path = os.path.join(ROOT, "tests", "golden", "cv2_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
