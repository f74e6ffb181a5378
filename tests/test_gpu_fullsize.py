"""GPU: the configurations BASELINE.json names, at their full sizes, against the CPU oracle.

Round-1 review: the edge term had only met the oracle on frames up to 131x257 with k <= 9, and the fused
HSV + luma + byte-sum variant on three 1080p frames.  Here: the reference's automatic dilation kernel at 1080p
(k = 13) and 4K (k = 19) (``content_detector.py:39-46, 213-239``), kernels whose windows straddle 32-bit words of
the bit rows (k = 31, 33, 63), a weak-edge chain that needs dozens of hysteresis relaunches, and the fused variant
on 64 uniform and on constant frames at 1080p.  Everything bit-exact.
"""
import numpy as np
import pytest

from oracle import lib as orc
from oracle.detectors_np import edge_map, estimated_kernel_size
from oracle.detectors_np import score_batch as oracle_score
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.synth import make_clip

pytestmark = pytest.mark.gpu
NOEDGE = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
FIELDS = ("sad_h", "sad_s", "sad_v", "byte_sum", "hist")


def same(a, b, fields=FIELDS):
    for f in fields:
        assert np.array_equal(a[f], b[f]), f"field {f} differs at {np.argwhere(a[f] != b[f])[:4].tolist()}"


def device_copy(engine, frames):
    buf = engine.alloc(frames.nbytes)
    buf.upload(frames.reshape(-1))
    return buf


def shots_with_objects(seed, n, h, w, noise=3.0):
    """Shot-like frames that actually contain edges: the smooth synthetic shots plus, per frame, a few flat
    rectangles and a diagonal band that move a little from frame to frame."""
    frames, _ = make_clip(seed, n, h, w, shot_len=(2, 3), noise=noise, fade_every=0)
    rng = np.random.default_rng(seed + 1)
    rects = [(int(rng.integers(0, h - h // 4)), int(rng.integers(0, w - w // 4)), int(rng.integers(h // 16, h // 4)),
              int(rng.integers(w // 16, w // 4)), rng.integers(0, 256, 3)) for _ in range(6)]
    yy, xx = np.mgrid[0:h, 0:w]
    for t in range(n):
        for (y, x, rh, rw, col) in rects:
            y2, x2 = min(h - rh, y + 3 * t), min(w - rw, x + 5 * t)
            frames[t, y2:y2 + rh, x2:x2 + rw] = col
        band = np.abs(yy - xx * h // w - 7 * t) < max(2, h // 90)
        frames[t][band] = (frames[t][band].astype(np.int32) + 80).clip(0, 255).astype(np.uint8)
    return frames


def check_edge_maps(engine, frames, k, which):
    n, h, w, _ = frames.shape
    buf = device_copy(engine, frames)
    for i in which:
        got = engine.edge_map(buf.ptr + i * h * w * 3, h, w, edge_kernel=k)
        want = edge_map(frames[i], k)
        assert np.array_equal(got, want), f"{w}x{h} k={k} frame {i}: {np.count_nonzero(got != want)} px differ"
    buf.free()


def test_edges_1080p_auto_kernel_shot_like_and_uniform(hip_engine):
    """BASELINE config 2, second run: weights (1,1,1,1) at 1920x1080, where the reference picks k = 13."""
    assert estimated_kernel_size(1920, 1080) == 13
    shots = shots_with_objects(41, 8, 1080, 1920)
    rng = np.random.default_rng(42)
    noise = rng.integers(0, 256, (3, 1080, 1920, 3), dtype=np.uint8)
    for frames, which in ((shots, (0, 5)), (noise, (1,))):
        want = oracle_score(frames, edges=True)
        buf = device_copy(hip_engine, frames)
        got = hip_engine.score_device(buf.ptr, len(frames), 1080, 1920, flags=E.SCORE_ALL)
        buf.free()
        same(got, want, FIELDS + ("edge_xor",))
        if frames is shots:      # (uniform noise dilated by 13x13 is all edge in every frame: its XOR counts are 0)
            assert want["edge_xor"][1:].all()
        check_edge_maps(hip_engine, frames, 0, which)


def test_edges_4k_auto_kernel(hip_engine):
    """3840x2160: k = 19."""
    assert estimated_kernel_size(3840, 2160) == 19
    shots = shots_with_objects(43, 2, 2160, 3840)
    rng = np.random.default_rng(44)
    frames = np.concatenate([shots, rng.integers(0, 256, (1, 2160, 3840, 3), dtype=np.uint8)])
    want = oracle_score(frames, edges=True)
    buf = device_copy(hip_engine, frames)
    got = hip_engine.score_device(buf.ptr, 3, 2160, 3840, flags=E.SCORE_ALL)
    buf.free()
    same(got, want, FIELDS + ("edge_xor",))
    check_edge_maps(hip_engine, frames, 0, (0, 2))


@pytest.mark.parametrize("k", [13, 19, 31, 33, 63])
def test_dilation_windows_across_word_boundaries(hip_engine, k):
    """Horizontal windows wider than a 32-bit word of the bit rows, vertical windows taller than the frame edge."""
    shots = shots_with_objects(50 + k, 3, 150, 260, noise=5.0)
    rng = np.random.default_rng(k)
    sparse = np.full((2, 97, 131, 3), 60, np.uint8)          # odd width: the last word of a bit row is partial
    sparse[0, 40:44, 60:64] = 250
    sparse[1, 3:5, 125:131] = 250                            # touches the right border
    sparse[1, 90:97, 0:3] = 250                              # and the bottom-left corner
    noise = rng.integers(0, 256, (2, 70, 100, 3), dtype=np.uint8)
    for frames in (shots, sparse, noise):
        n, h, w, _ = frames.shape
        check_edge_maps(hip_engine, frames, k, range(n))
        got = hip_engine.score_host(frames, flags=E.SCORE_EDGES, edge_kernel=k)
        want = oracle_score(frames, edges=True, kernel_size=k)
        assert np.array_equal(got["edge_xor"], want["edge_xor"])


def serpentine(h, w, seed=True, m=90, s=20, strong=70, pitch=16, width=8):
    """A corridor 20 grey levels above the background that snakes across the frame: its outline is ONE connected chain
    of weak Canny pixels (4 * 20 = 80 between the thresholds 60 and 120 that the median 90 gives); a small bright
    patch at the start of the corridor supplies the only strong pixels."""
    img = np.full((h, w), m, np.uint8)
    runs = list(range(8, w - 8 - width, pitch))
    for i, x0 in enumerate(runs):
        img[8:h - 8, x0:x0 + width] = m + s
        if i + 1 < len(runs):
            x1 = runs[i + 1]
            if i % 2 == 0:
                img[h - 8 - width:h - 8, x0:x1 + width] = m + s
            else:
                img[8:8 + width, x0:x1 + width] = m + s
    if seed:
        img[20:26, runs[0]:runs[0] + 4] = m + s + strong
    return np.repeat(img[:, :, None], 3, axis=2), runs


@pytest.mark.parametrize("shape", [(512, 512), (300, 700), (1080, 1920)])
def test_hysteresis_chain_far_longer_than_one_round_of_launches(hip_engine, shape):
    """The chain crosses hundreds of 64x64 hysteresis tiles one after the other, so the device has to relaunch
    the hysteresis kernel dozens of times; without the bright patch there is no edge at all."""
    h, w = shape
    frame, runs = serpentine(h, w)
    want = edge_map(frame, 3)
    assert not edge_map(serpentine(h, w, seed=False)[0], 3).any()
    assert want[:, runs[-1] - 4:runs[-1] + 12].any(), "the chain must reach the far end of the corridor"
    buf = device_copy(hip_engine, frame[None])
    got = hip_engine.edge_map(buf.ptr, h, w, edge_kernel=3)
    buf.free()
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} px differ"
    # as the second frame of a batch: the XOR count against the seedless frame is the whole chain
    pair = np.stack([serpentine(h, w, seed=False)[0], frame])
    rec = hip_engine.score_host(pair, flags=E.SCORE_EDGES, edge_kernel=3)
    assert rec["edge_xor"][1] == np.count_nonzero(want)


def test_fused_all_terms_1080p_64_uniform_frames(hip_engine):
    """BASELINE config 5's kernel (HSV + luma histogram + byte sum in one pass) on 64 full-size frames."""
    rng = np.random.default_rng(64)
    frames = rng.integers(0, 256, (64, 1080, 1920, 3), dtype=np.uint8)
    prev = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    want = orc.score_batch(frames, prev)
    buf = device_copy(hip_engine, np.concatenate([prev[None], frames]))
    stride = 1080 * 1920 * 3
    got = hip_engine.score_device(buf.ptr + stride, 64, 1080, 1920, d_prev=buf.ptr, flags=NOEDGE)
    same(got, want)
    same(hip_engine.score_device(buf.ptr + stride, 64, 1080, 1920, d_prev=buf.ptr, flags=E.SCORE_HSV_SAD), want,
         ("sad_h", "sad_s", "sad_v"))
    same(hip_engine.score_device(buf.ptr + stride, 64, 1080, 1920, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM), want,
         ("hist", "byte_sum"))
    buf.free()


def test_fused_all_terms_1080p_constant_frames(hip_engine):
    """The K distribution (every pixel of a frame in ONE histogram bin) at full size, fused and separate passes."""
    vals = [(0, 0, 0), (255, 255, 255), (128, 128, 128), (255, 0, 0), (0, 255, 0), (0, 0, 255), (17, 200, 99), (1, 1, 2),
            (254, 255, 254), (128, 128, 128), (128, 128, 128), (90, 10, 200)]
    frames = np.empty((len(vals), 1080, 1920, 3), np.uint8)
    for i, v in enumerate(vals):
        frames[i] = v
    want = orc.score_batch(frames)
    buf = device_copy(hip_engine, frames)
    same(hip_engine.score_device(buf.ptr, len(vals), 1080, 1920, flags=NOEDGE), want)
    same(hip_engine.score_device(buf.ptr, len(vals), 1080, 1920, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM), want, ("hist", "byte_sum"))
    same(hip_engine.score_device(buf.ptr, len(vals), 1080, 1920, flags=E.SCORE_HSV_SAD), want, ("sad_h", "sad_s", "sad_v"))
    buf.free()


# ---- the reference's default pipeline: downscale, then score (scene_manager.py:666-678) ---------------------------

def oracle_downscaled(frames, prev, dh, dw, interpolation=1):
    import cv2  # the oracle shim

    small = np.stack([cv2.resize(f, (dw, dh), interpolation=interpolation) for f in frames])
    sprev = cv2.resize(prev, (dw, dh), interpolation=interpolation) if prev is not None else None
    return small, orc.score_batch(small, sprev)


@pytest.mark.parametrize("case", [
    (70, 180, 320, 144, 256),      # scale 1.25: neighbouring destination rows share source rows; 3 time chunks
    (9, 1080, 1920, 144, 256),     # the default target of a 1080p source (factor 7.5)
    (5, 2160, 3840, 144, 256),     # 4K source: one destination row per workgroup
    (40, 720, 1280, 144, 256),
    (12, 200, 304, 100, 152),      # exact 2x2 decimation: OpenCV's INTER_AREA shortcut
    (6, 97, 131, 41, 77),          # odd sizes: rows not 16-byte aligned -> resize, then score
    (33, 360, 640, 203, 361),      # wide destination rows (several pixels per thread)
])
def test_downscale_then_score_matches_oracle(hip_engine, case):
    n, sh, sw, dh, dw = case
    rng = np.random.default_rng(n * sh)
    frames = rng.integers(0, 256, (n, sh, sw, 3), dtype=np.uint8)
    if n > 8:   # shot-like stretches so that consecutive frames are close (small SADs) as well as far apart
        frames[3:8] = frames[3]
        frames[5, sh // 3: sh // 2] ^= 7
    prev = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    stride = (sh * sw * 3 + 15) & ~15
    buf = hip_engine.alloc(stride * (n + 1))
    buf.upload(prev.reshape(-1), 0)
    for i in range(n):
        buf.upload(frames[i].reshape(-1), (i + 1) * stride)
    for with_prev in (False, True):
        _, want = oracle_downscaled(frames, prev if with_prev else None, dh, dw)
        # every set of non-edge terms takes ONE fused kernel (round 4: the luma histogram and the byte sum too)
        for flags, fields in ((E.SCORE_HSV_SAD, ("sad_h", "sad_s", "sad_v")), (NOEDGE, FIELDS),
                              (E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM, ("hist", "byte_sum")), (E.SCORE_BYTE_SUM, ("byte_sum",)),
                              (E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST, ("sad_h", "sad_s", "sad_v", "hist"))):
            got = hip_engine.score_device_downscaled(buf.ptr + stride, n, sh, sw, dh, dw, frame_stride=stride,
                                                     d_prev=buf.ptr if with_prev else None, flags=flags)
            same(got, want, fields)
            assert not got["edge_xor"].any()
            if not flags & E.SCORE_HSV_SAD:
                assert not got["sad_h"].any() and not got["sad_v"].any()
    buf.free()


def test_downscale_then_all_four_constant_frames_and_headline_size(hip_engine):
    """The fused downscale + HSV + luma + byte-sum kernel (BASELINE configs[4]'s detector set behind the reference's default
    downscale, scene_manager.py:666-678): constant frames (every pixel of a tile in ONE histogram bin) against the oracle, and
    4096 x 1080p -> 256 x 144: chunking invariance (one call == four calls with a predecessor), histograms that sum to the small
    frame, and the oracle on both sides of walk boundaries deep in the batch."""
    import torch

    vals = [(0, 0, 0), (255, 255, 255), (128, 128, 128), (255, 0, 0), (0, 255, 0), (0, 0, 255), (17, 200, 99), (128, 128, 128), (90, 10, 200)]
    frames = np.empty((len(vals), 360, 640, 3), np.uint8)
    for i, v in enumerate(vals):
        frames[i] = v
    buf = device_copy(hip_engine, frames)
    _, want = oracle_downscaled(frames, None, 144, 256)
    same(hip_engine.score_device_downscaled(buf.ptr, len(vals), 360, 640, 144, 256, flags=NOEDGE), want)
    buf.free()
    n, sh, sw, dh, dw = 4096, 1080, 1920, 144, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(78)
    x = torch.empty((n, sh, sw, 3), dtype=torch.uint8, device="cuda")
    for i in range(0, n, 64):
        x[i:i + 64] = torch.randint(0, 256, (64, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    x[900:1000] //= 8                        # dark frames: other histogram bins, a fade for ThresholdDetector
    torch.cuda.synchronize()
    stride = sh * sw * 3
    whole = hip_engine.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=NOEDGE)
    walk, _ = hip_engine.last_walk_geometry()
    assert (whole["hist"].sum(axis=1) == dh * dw).all() and whole["byte_sum"][950] < whole["byte_sum"][10] // 4
    hsv_only = hip_engine.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=E.SCORE_HSV_SAD)
    same(whole, hsv_only, ("sad_h", "sad_s", "sad_v"))
    luma_only = hip_engine.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM)
    same(whole, luma_only, ("hist", "byte_sum"))
    parts = [hip_engine.score_device_downscaled(x.data_ptr() + a * stride, 1024, sh, sw, dh, dw, flags=NOEDGE,
                                                d_prev=x.data_ptr() + (a - 1) * stride if a else None) for a in range(0, n, 1024)]
    same(np.concatenate(parts), whole)
    assert 8 <= walk < n
    for a in (0, walk - 1, (n // 2 // walk) * walk - 1, n - 3):
        _, want = oracle_downscaled(x[a:a + 3].cpu().numpy(), x[a - 1].cpu().numpy() if a else None, dh, dw)
        same(whole[a:a + 3], want)


@pytest.mark.parametrize("shape", [(180, 320, 144, 256), (360, 640, 144, 256), (1080, 1920, 144, 256), (2160, 3840, 144, 256), (360, 640, 203, 361)])
def test_partial_histograms_whose_counts_pass_a_byte(hip_engine, shape):
    """The fused downscale kernel hands a tile's luma histogram over as the low bytes of its counts plus escapes for the bins a byte cannot
    hold (rs_hist_flush).  Frames that fill the escapes: constant ones (every pixel of a tile in ONE bin: 2048 at 320 x 180 -> 256 x 144, the
    largest tile there is), two and eight flat bands (bins of exactly 256, 512 ... per tile; several escape lanes at once), a flat frame with a
    sprinkle of noise (counts of 255 / 256 / 257 side by side), next to plain noise -- histograms, byte sums and SADs against the oracle;
    the V-histogram instance of the same machinery through the edge term."""
    sh, sw, dh, dw = shape
    rng = np.random.default_rng(sh + dw)
    frames = []
    for v in (0, 255, 128, 17):
        frames.append(np.full((sh, sw, 3), v, np.uint8))
    two = np.zeros((sh, sw, 3), np.uint8)
    two[:, sw // 2:] = (200, 90, 30)
    frames.append(two)
    bands = np.zeros((sh, sw, 3), np.uint8)
    for k in range(8):
        bands[:, k * sw // 8:(k + 1) * sw // 8] = (31 * k, 255 - 29 * k, 7 * k)
    frames.append(bands)
    rows = np.zeros((sh, sw, 3), np.uint8)
    for k in range(sh):
        rows[k] = (k * 7) % 256                      # (every source row its own grey: the tiles of a frame differ)
    frames.append(rows)
    for density in (0.001, 0.01, 0.4):
        f = np.full((sh, sw, 3), 90, np.uint8)
        m = rng.random((sh, sw)) < density
        f[m] = rng.integers(0, 256, (int(m.sum()), 3), dtype=np.uint8)
        frames.append(f)
    frames.append(rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8))
    frames = np.stack(frames + frames[::-1])
    buf = device_copy(hip_engine, frames)
    small, want = oracle_downscaled(frames, None, dh, dw)
    assert want["hist"].max() == dh * dw
    for flags in (NOEDGE, E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM):
        got = hip_engine.score_device_downscaled(buf.ptr, len(frames), sh, sw, dh, dw, flags=flags)
        same(got, want, FIELDS if flags == NOEDGE else ("hist", "byte_sum"))
    if dw == 256:
        got = hip_engine.score_device_downscaled(buf.ptr, len(frames), sh, sw, dh, dw, flags=E.SCORE_HSV_SAD | E.SCORE_EDGES)
        same(got, oracle_score(small, edges=True), ("sad_h", "sad_s", "sad_v", "edge_xor"))
    buf.free()


@pytest.mark.parametrize("shape", [(1080, 1920, 540, 960), (360, 640, 203, 361), (1080, 1920, 144, 256), (2160, 3840, 1080, 1920)])
def test_v_histograms_of_the_fused_front_end_from_its_planes(hip_engine, shape):
    """HSV + edge term behind a downscale: the V histograms (numpy.median for Canny's thresholds, content_detector.py:229-233) are counted
    from the V planes the fused downscale kernel wrote (vplane_hist_kernel): one workgroup per frame at the default size, several with
    global atomics above 256 K pixels (960 x 540, 1920 x 1080), byte loads where a plane does not start on 16 bytes (361 x 203: odd
    pixel count).  Flat frames, bands and noise -- medians at the ends, between two bins, in the middle -- against the oracle."""
    sh, sw, dh, dw = shape
    rng = np.random.default_rng(sh * 7 + dw)
    frames = [np.full((sh, sw, 3), v, np.uint8) for v in (0, 255, 90)]
    two = np.zeros((sh, sw, 3), np.uint8)
    two[:, sw // 2:] = (200, 90, 30)                        # half the pixels V = 0, half V = 200: the median lies between two bins
    frames.append(two)
    box = rng.integers(60, 70, (sh, sw, 3), dtype=np.uint8)
    box[sh // 4:3 * sh // 4, sw // 3:2 * sw // 3] = (250, 10, 10)
    frames.append(box)
    frames.append(rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8))
    frames = np.stack(frames + frames[::-1])
    buf = device_copy(hip_engine, frames)
    small, _ = oracle_downscaled(frames, None, dh, dw)
    got = hip_engine.score_device_downscaled(buf.ptr, len(frames), sh, sw, dh, dw, flags=E.SCORE_HSV_SAD | E.SCORE_EDGES)
    same(got, oracle_score(small, edges=True), ("sad_h", "sad_s", "sad_v", "edge_xor"))
    buf.free()


def test_downscale_then_score_other_terms_and_modes(hip_engine):
    """Edges and the NEAREST / AREA / LANCZOS4 / CUBIC modes go through the resize-then-score path of the same entry point."""
    import cv2  # the oracle shim

    n, sh, sw, dh, dw = 5, 360, 640, 144, 256
    frames = shots_with_objects(77, n, sh, sw)
    buf = device_copy(hip_engine, frames)
    for interpolation in (1, 0, 3, 4, 2):       # (cv2's values: LINEAR, NEAREST, AREA, LANCZOS4, CUBIC)
        small = np.stack([cv2.resize(f, (dw, dh), interpolation=interpolation) for f in frames])
        want = oracle_score(small, edges=True)
        got = hip_engine.score_device_downscaled(buf.ptr, n, sh, sw, dh, dw, flags=E.SCORE_ALL, interpolation=interpolation)
        same(got, want, FIELDS + ("edge_xor",))
    with pytest.raises(NotImplementedError):
        hip_engine.score_device_downscaled(buf.ptr, n, sh, sw, dh, dw, interpolation=5)
    buf.free()


def test_downscaled_submissions_pipeline(hip_engine):
    import torch

    n, sh, sw, dh, dw = 64, 1080, 1920, 144, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    x = torch.randint(0, 256, (n, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    torch.cuda.synchronize()
    hip_engine.submit_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw)
    hip_engine.submit_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw)
    a, b = hip_engine.collect(n), hip_engine.collect(n)
    same(a, b, ("sad_h", "sad_s", "sad_v"))
    _, want = oracle_downscaled(x[:6].cpu().numpy(), None, dh, dw)
    same(a[:6], want, ("sad_h", "sad_s", "sad_v"))


# ---- many clips packed into one batch (north_star; psd_score_segments_device) ----------------------------------------

def test_packed_clips_equal_per_clip_records(hip_engine):
    rng = np.random.default_rng(99)
    shapes = [(7, 72, 128), (1, 72, 128), (33, 72, 128), (5, 90, 160), (2, 72, 128), (12, 90, 160), (70, 72, 128), (3, 37, 53)]
    clips = [rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8) for n, h, w in shapes]
    for flags, fields in ((NOEDGE, FIELDS), (E.SCORE_HSV_SAD, ("sad_h", "sad_s", "sad_v")), (E.SCORE_ALL, FIELDS + ("edge_xor",))):
        want = [oracle_score(c, edges=bool(flags & E.SCORE_EDGES)) for c in clips]
        got = hip_engine.score_clips(clips, flags=flags)
        for g, wv, shp in zip(got, want, shapes):
            assert len(g) == shp[0]
            same(g, wv, fields)
        # tiny batches: clips spill over several launches
        got = hip_engine.score_clips(clips, flags=flags, max_batch_bytes=40 * 72 * 128 * 3)
        for g, wv in zip(got, want):
            same(g, wv, fields)


def test_packed_device_clips_and_segment_validation(hip_engine):
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    big = torch.randint(0, 256, (96, 180, 320, 3), dtype=torch.uint8, device="cuda", generator=g)
    torch.cuda.synchronize()
    lens = [10, 1, 40, 45]
    clips, off = [], 0
    for n in lens:
        clips.append(big[off:off + n])
        off += n
    got = hip_engine.score_clips(clips, flags=NOEDGE)
    host = big.cpu().numpy()
    off = 0
    for gr, n in zip(got, lens):
        same(gr, orc.score_batch(host[off:off + n]))
        off += n
    # one launch over the whole tensor with the segment table == the per-clip results
    recs = hip_engine.score_device_segments(big.data_ptr(), 96, 180, 320, [0, 10, 11, 51], flags=NOEDGE)
    same(recs, np.concatenate(got))
    with pytest.raises(ValueError):
        hip_engine.score_device_segments(big.data_ptr(), 96, 180, 320, [0, 11, 10], flags=NOEDGE)
    with pytest.raises(ValueError):
        hip_engine.score_device_segments(big.data_ptr(), 96, 180, 320, [0, 96], flags=NOEDGE)


# ---- the exchange step through the C-ABI (psd_allgather_scores, RCCL loaded by libpsd_hip.so) -----------------------------

def test_native_rccl_all_gather_of_device_records_one_rank(hip_engine):
    """One GPU per box here, so one rank: the communicator comes up, the records travel device -> RCCL -> host and come
    back identical; a count that disagrees with the contribution is refused.  (N > 1 runs on the driver's 8-GPU node
    through the same code; the gloo tests cover the host logic around it.)"""
    from pyscenedetect_amd.distributed import NativeComm

    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (37, 72, 128, 3), dtype=np.uint8)
    want = hip_engine.score_host(frames, flags=NOEDGE)
    buf = device_copy(hip_engine, frames)
    recs = hip_engine.score_device(buf.ptr, 37, 72, 128, flags=NOEDGE)
    same(recs, want)
    comm = NativeComm(hip_engine, 1, 0, NativeComm.make_unique_id())
    parts = comm.all_gather_records([37])
    assert len(parts) == 1 and parts[0].tobytes() == recs.tobytes()
    recs2 = hip_engine.score_device(buf.ptr, 5, 72, 128, flags=E.SCORE_HSV_SAD)
    assert comm.all_gather_records([5], local=recs2)[0].tobytes() == recs2.tobytes()      # smaller than the buffers: reuse
    with pytest.raises(ValueError):
        comm.all_gather_records([6])
    # records a rank holds on the host (psd_allgather_host): both kinds, an empty contribution, a count that disagrees
    for local in (recs, E._sums_of(recs), recs[:0]):
        got = comm.all_gather_host(local, [len(local)])
        assert len(got) == 1 and got[0].dtype == local.dtype and got[0].tobytes() == local.tobytes()
    with pytest.raises(ValueError):
        comm.all_gather_host(recs[:4], [5])
    assert comm.all_gather_host(recs2, [5])[0].tobytes() == recs2.tobytes()              # ... and the communicator is still usable
    comm.close()
    buf.free()


# ---- threads: every thread gets its own default engine; SceneManagers in parallel -------------------------------------------

def test_scene_managers_on_two_threads_use_their_own_default_engines(golden):
    import threading

    import pyscenedetect_amd as psd
    from pyscenedetect_amd.engine import default_engine
    from tests.conftest import golden_clip

    jobs = [("wide_d", "content_default", psd.ContentDetector, True), ("scenes_a", "hist_default", psd.HistogramDetector, False)]
    engines, errors = {}, []

    def work(i):
        clip, cfg, cls, auto = jobs[i]
        try:
            engines[i] = default_engine()
            frames = golden_clip(golden, clip)
            for _ in range(3):
                sm = psd.SceneManager(batch_frames=16)          # no engine given: the thread's default engine
                sm.auto_downscale = auto
                sm.add_detector(cls())
                sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
                assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == golden["clips"][clip]["results"][cfg]["cuts"]
        except Exception as ex:  # noqa: BLE001
            errors.append((i, ex))

    for name in ("wide_d", "scenes_a"):
        golden_clip(golden, name)      # build the shared cache on this thread first
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert engines[0] is not engines[1] and default_engine() is not engines[0]


def test_downscaled_scoring_properties_at_the_headline_size(hip_engine):
    """4096 x 1080p behind the default downscale: chunking invariance (one call == four calls with a predecessor frame),
    idempotence, and an oracle sample at a chunk boundary of the kernel's time walk."""
    import torch

    n, sh, sw, dh, dw = 4096, 1080, 1920, 144, 256
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    x = torch.empty((n, sh, sw, 3), dtype=torch.uint8, device="cuda")
    for i in range(0, n, 64):
        x[i:i + 64] = torch.randint(0, 256, (64, sh, sw, 3), dtype=torch.uint8, device="cuda", generator=g)
    x[100:140] = x[100]                      # a run of identical frames: zero SADs
    torch.cuda.synchronize()
    stride = sh * sw * 3
    whole = hip_engine.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw)
    again = hip_engine.score_device_downscaled(x.data_ptr(), n, sh, sw, dh, dw)
    same(whole, again, ("sad_h", "sad_s", "sad_v"))
    assert not whole["sad_v"][101:140].any() and whole["sad_v"][100] > 0 and whole["sad_v"][140] > 0 and whole["sad_h"][0] == 0
    parts = []
    for a in range(0, n, 1024):
        parts.append(hip_engine.score_device_downscaled(x.data_ptr() + a * stride, 1024, sh, sw, dh, dw,
                                                        d_prev=x.data_ptr() + (a - 1) * stride if a else None))
    same(np.concatenate(parts), whole, ("sad_h", "sad_s", "sad_v"))
    for a in (30, 63, 2047):                 # around the ends of 32-frame walks
        _, want = oracle_downscaled(x[a:a + 3].cpu().numpy(), x[a - 1].cpu().numpy(), dh, dw)
        same(whole[a:a + 3], want, ("sad_h", "sad_s", "sad_v"))


def test_hsv_and_edges_from_one_read_across_workspace_chunks():
    """HSV + edge terms together take the V-mode HSV pass (one read of the frames).  With a 1 MiB edge workspace the batch
    is cut into several chunks: the first starts with the predecessor frame as virtual frame 0, the later ones chain on
    the frame in front of them.  Must equal the oracle and the two-read path (PSD_EDGE_FUSE_HSV=0).  Subprocess: the
    environment switches are read once per process."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
from oracle.detectors_np import score_batch as oracle_score
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.synth import make_clip
frames, _ = make_clip(31, 121, 72, 128, shot_len=(9, 17))
eng = E.ScoringEngine(0)
want = oracle_score(frames, edges=True)
fields = ("sad_h", "sad_s", "sad_v", "edge_xor")
flags = E.SCORE_HSV_SAD | E.SCORE_EDGES
got = eng.score_host(frames[1:], prev=frames[0], flags=flags)
for f in fields:
    assert np.array_equal(got[f], want[f][1:]), f
got = eng.score_host(frames, flags=E.SCORE_ALL)
for f in fields + ("byte_sum",):
    assert np.array_equal(got[f], want[f]), f
assert np.array_equal(got["hist"], want["hist"])
print("ok")
""" % root
    for fuse in ("1", "0"):
        env = dict(os.environ, PSD_EDGE_WS_MB="1", PSD_EDGE_FUSE_HSV=fuse)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (fuse, out.stdout[-400:], out.stderr[-1200:])


def test_a_scene_manager_outlives_the_thread_that_built_it(golden):
    """Advisor, round 2: default_engine() used to destroy the engines of dead threads, so a SceneManager first used on a
    worker thread lost its engine as soon as any other thread asked for a new one.  The engine now lives as long as
    something holds it."""
    import threading

    import pyscenedetect_amd as psd
    from pyscenedetect_amd.engine import default_engine
    from tests.conftest import golden_clip

    frames = golden_clip(golden, "scenes_a")
    want = golden["clips"]["scenes_a"]["results"]["content_default"]["cuts"]
    box = {}

    def build():
        sm = psd.SceneManager(batch_frames=16)
        sm.auto_downscale = False
        sm.add_detector(psd.ContentDetector())
        sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
        box["sm"], box["engine"] = sm, default_engine()

    t = threading.Thread(target=build)
    t.start()
    t.join()
    other = threading.Thread(target=lambda: box.setdefault("other", default_engine()))   # a NEW engine on a new thread
    other.start()
    other.join()
    assert box["engine"]._h is not None and box["other"] is not box["engine"]
    sm = box["sm"]
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == want
    sm2 = psd.SceneManager(engine=box["engine"], batch_frames=16)       # and the dead thread's engine still scores
    sm2.auto_downscale = False
    sm2.add_detector(psd.ContentDetector(engine=box["engine"]))
    sm2.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    assert [c.frame_num for c in sm2.get_cut_list(show_warning=False)] == want


# ---- ABI 3: records without the histogram (psd_frame_sums, psd_score_collect_sums) ------------------------------------------


def test_collect_sums_equals_the_heads_of_full_records(hip_engine):
    """Submissions without the luma terms move only 40 bytes per frame to the host (one strided device -> host copy);
    whatever was submitted, collect(sums_only=True) is the head of what collect() returns, and the full form keeps its
    zero histogram."""
    import torch
    from pyscenedetect_amd._native import SUMS_DTYPE

    g = torch.Generator(device="cuda")
    g.manual_seed(31)
    x = torch.randint(0, 256, (70, 144, 256, 3), dtype=torch.uint8, device="cuda", generator=g)
    torch.cuda.synchronize()
    want = oracle_score(x.cpu().numpy(), edges=True)
    for flags in (E.SCORE_HSV_SAD, E.SCORE_HSV_SAD | E.SCORE_EDGES, E.SCORE_EDGES, E.SCORE_BYTE_SUM, E.SCORE_ALL, NOEDGE):
        fields = ([f for f in ("sad_h", "sad_s", "sad_v") if flags & E.SCORE_HSV_SAD] + (["edge_xor"] if flags & E.SCORE_EDGES else []) +
                  (["byte_sum"] if flags & (E.SCORE_BYTE_SUM | E.SCORE_LUMA_HIST) else []))
        full = hip_engine.score_device(x.data_ptr(), 70, 144, 256, flags=flags)
        sums = hip_engine.score_device(x.data_ptr(), 70, 144, 256, flags=flags, sums_only=True)
        assert full.dtype == E.RECORD_DTYPE and sums.dtype == SUMS_DTYPE
        for f in SUMS_DTYPE.names:
            assert np.array_equal(full[f], sums[f]), (flags, f)
        same(sums, want, fields)
        if flags & (E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM):
            assert np.array_equal(full["hist"], want["hist"]), flags
        else:
            assert not full["hist"].any(), flags
    # both forms through the pipelined API, interleaved, sizes differing (the pinned mirror of a slot is reused)
    for n, so in ((70, True), (3, False), (64, True), (70, False)):
        hip_engine.submit_device(x.data_ptr(), n, 144, 256, flags=E.SCORE_HSV_SAD if n != 3 else NOEDGE)
    for n, so in ((70, True), (3, False), (64, True), (70, False)):
        r = hip_engine.collect(n, sums_only=so)
        same(r, want[:n], ("sad_h", "sad_s", "sad_v"))
    with pytest.raises(ValueError):
        hip_engine.collect(5, sums_only=True)      # nothing pending


def test_score_clips_pipelines_more_jobs_than_slots(hip_engine):
    """Six runs of resident clips (alternating resolutions, so no two neighbours are contiguous): more submissions than
    PSD_MAX_INFLIGHT, collected in windows; with and without the histogram."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    shapes = [(9, 72, 128), (4, 90, 160), (17, 72, 128), (6, 90, 160), (30, 72, 128), (1, 90, 160), (12, 36, 64)]
    pool = {}
    clips = []
    for n, h, w in shapes:
        # separate allocations with a gap: runs stay apart
        t = torch.randint(0, 256, (n + 1, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
        pool[len(clips)] = t
        clips.append(t[:n])
    # ... and one run of three clips back to back
    big = torch.randint(0, 256, (20, 72, 128, 3), dtype=torch.uint8, device="cuda", generator=g)
    clips += [big[:5], big[5:6], big[6:]]
    torch.cuda.synchronize()
    want = [orc.score_batch(c.cpu().numpy()) for c in clips]
    got = hip_engine.score_clips(clips, flags=NOEDGE)
    for gr, wv in zip(got, want):
        same(gr, wv)
    sums = hip_engine.score_clips(clips, flags=E.SCORE_HSV_SAD, sums_only=True)
    for gr, wv in zip(sums, want):
        assert gr.dtype.itemsize == 40 and len(gr) == len(wv)
        same(gr, wv, ("sad_h", "sad_s", "sad_v"))
    host = [c.cpu().numpy() for c in clips[:3]]
    for gr, wv in zip(hip_engine.score_clips(host, flags=E.SCORE_HSV_SAD | E.SCORE_BYTE_SUM, sums_only=True), want):
        assert gr.dtype.itemsize == 40
        same(gr, wv, ("sad_h", "sad_s", "sad_v", "byte_sum"))


def test_score_clips_leaves_the_engine_usable_after_an_error(hip_engine):
    """An exception out of the caller's on_ready (or a refused submission) must not strand submissions in the engine's ring:
    synchronous calls refuse to run over pending ones."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    clips = [torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
             for n, h, w in ((5, 72, 128), (4, 90, 160), (6, 36, 64))]
    torch.cuda.synchronize()

    def boom(i, recs):
        raise RuntimeError("caller's callback failed")

    with pytest.raises(RuntimeError, match="callback failed"):
        hip_engine.score_clips(clips, flags=E.SCORE_HSV_SAD, on_ready=boom)
    # a bad clip among good ones: the submissions before it are retired
    bad = clips[:2] + [torch.zeros((3, 0, 8, 3), dtype=torch.uint8, device="cuda")]
    with pytest.raises(ValueError):
        hip_engine.score_clips(bad, flags=E.SCORE_HSV_SAD)
    got = hip_engine.score_clips(clips, flags=E.SCORE_HSV_SAD)
    for gr, c in zip(got, clips):
        same(gr, orc.score_batch(c.cpu().numpy()), ("sad_h", "sad_s", "sad_v"))
