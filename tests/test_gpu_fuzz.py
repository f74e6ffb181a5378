"""A fixed-seed slice of tools/fuzz_gpu.py: random frame shapes (single rows / columns, pixel counts around multiples of 16,
widths around the tile sizes), contents (noise, flat, grey, two-level, saturated primaries, dark), memory layouts (padded
rows / frames, misaligned bases, strided frames), term sets and entry points (score_host, score_frames, downscale with every
interpolation mode, score_clips with host / device / packed device clips -- since round 6 also behind the reference's default
downscale and every other resize setting (psd_score_segments_downscaled_device) --, the edge term with fixed and automatic dilation
sizes, hash thumbnails) through the HIP engine, records identical to the CPU oracle.  The long runs (20 k cases without a
mismatch, `profiles/r05_*_fuzz*.json`) are `python tools/fuzz_gpu.py --seconds 100`."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [20250922, 7])
def test_fuzz_slice_equals_the_oracle(hip_engine, seed):
    import fuzz_gpu

    # (bounded by cases, not by time: the first `import torch` of a fresh box alone can take a minute)
    out = fuzz_gpu.run(seed, seconds=300.0, max_cases=700, engine=hip_engine)
    assert out["cases"] == 700, out
    assert not out["mismatches"], out["mismatches"][:3]
    # every entry point was reached
    for entry in ("score_host", "score_frames", "downscale", "downscale/score_frames", "clips", "clips/downscaled", "edges", "hash"):
        assert out["by_entry"].get(entry, 0) > 0, out["by_entry"]
