"""GPU: the HIP scoring path (through the C-ABI) against the CPU oracle and the reference's
golden runs.  Integer records must be bit-exact; float metrics come out bit-identical because
the epilogue is the same arithmetic on identical integers (north_star tolerance for content_val
is 1e-4; we assert equality)."""
import numpy as np
import pytest

import pyscenedetect_amd as psd
from oracle import lib as orc
from oracle.detectors_np import score_batch as oracle_score
from pyscenedetect_amd import engine as E
from tests._helpers import assert_same_run, run_config
from tests.conftest import golden_clip

pytestmark = pytest.mark.gpu
NOEDGE = E.SCORE_HSV_SAD | E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM
FIELDS = ("sad_h", "sad_s", "sad_v", "byte_sum", "hist")


def device_copy(engine, frames):
    """frames uint8[N,H,W,3] copied into a fresh device buffer of the engine"""
    buf = engine.alloc(frames.nbytes)
    buf.upload(np.ascontiguousarray(frames).reshape(-1))
    return buf


def same(a, b, fields=FIELDS):
    for f in fields:
        assert np.array_equal(a[f], b[f]), f"field {f} differs: {np.argwhere(a[f] != b[f])[:4].tolist()}"


SHAPES = [(5, 36, 64), (7, 37, 53), (3, 144, 256), (4, 1, 1), (1, 1, 17), (9, 16, 16), (33, 90, 160), (2, 720, 1280),
          (130, 24, 40)]


@pytest.mark.parametrize("shape", SHAPES)
def test_records_match_oracle_host_and_device(hip_engine, shape):
    n, h, w = shape
    rng = np.random.default_rng(n * 1000 + h)
    fr = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    pv = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = orc.score_batch(fr)
    same(hip_engine.score_host(fr, flags=NOEDGE), ref)
    same(hip_engine.score_host(fr, prev=pv, flags=NOEDGE), orc.score_batch(fr, pv))
    buf = hip_engine.alloc(fr.nbytes + 16)
    buf.upload(fr.reshape(-1))
    same(hip_engine.score_device(buf.ptr, n, h, w, flags=NOEDGE), ref)           # packed (maybe unaligned)
    if fr.nbytes > 3:
        buf2 = hip_engine.alloc(fr.nbytes + 32)
        buf2.upload(fr.reshape(-1), offset=1)                                     # misaligned base pointer
        same(hip_engine.score_device(buf2.ptr + 1, n, h, w, flags=NOEDGE), ref)
    r = hip_engine.score_device(buf.ptr, n, h, w, flags=E.SCORE_HSV_SAD)
    same(r, ref, ("sad_h", "sad_s", "sad_v"))
    assert not r["hist"].any() and not r["byte_sum"].any()
    r = hip_engine.score_device(buf.ptr, n, h, w, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM)
    same(r, ref, ("hist", "byte_sum"))
    assert not r["sad_h"].any()


def test_padded_rows_and_frames(hip_engine):
    rng = np.random.default_rng(3)
    n, h, w = 6, 21, 45
    big = rng.integers(0, 256, (n, h + 3, w + 5, 3), dtype=np.uint8)
    view = big[:, 1:h + 1, 2:w + 2]          # non-contiguous rows and frames
    ref = orc.score_batch(np.ascontiguousarray(view))
    same(hip_engine.score_host(view, flags=NOEDGE), ref)


def test_exhaustive_hsv_through_the_kernel(hip_engine):
    """All 2^24 BGR triples as frame 1 against black frame 0: SAD sums = sums of H,S,V planes."""
    g = np.arange(256, dtype=np.uint8)
    cube = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(4096, 4096, 3)
    fr = np.stack([np.zeros_like(cube), cube])
    got = hip_engine.score_host(fr, flags=NOEDGE)
    same(got, orc.score_batch(fr))
    # row-wise too, so a compensating error inside the big sum cannot hide
    rows = cube.reshape(64, 64, 4096, 3)[:, 0]     # 64 slices of 4096 px
    fr2 = np.stack([np.zeros_like(rows[:, None]), rows[:, None]], axis=1).reshape(128, 1, 4096, 3)
    want2 = orc.score_batch(fr2)
    same(hip_engine.score_host(fr2, flags=NOEDGE), want2)
    # the HSV-only pass is its own kernel instantiation (fp32 formulation, single sdiv table): the same triples through it
    sad = ("sad_h", "sad_s", "sad_v")
    same(hip_engine.score_host(fr, flags=E.SCORE_HSV_SAD), got, sad)
    same(hip_engine.score_host(fr2, flags=E.SCORE_HSV_SAD), want2, sad)


def test_constant_and_extreme_frames(hip_engine):
    for val in (0, 1, 127, 128, 254, 255):
        fr = np.full((3, 72, 128, 3), val, np.uint8)
        same(hip_engine.score_host(fr, flags=NOEDGE), orc.score_batch(fr))
    fr = np.zeros((4, 40, 64, 3), np.uint8)
    fr[1] = 255
    fr[2, :, :, 2] = 255
    fr[3, :, :, 0] = 255
    same(hip_engine.score_host(fr, flags=NOEDGE), orc.score_batch(fr))


def test_empty_batch(hip_engine):
    r = hip_engine.score_host(np.zeros((0, 8, 8, 3), np.uint8), flags=NOEDGE)
    assert len(r) == 0


def test_invalid_arguments(hip_engine):
    with pytest.raises(ValueError):
        hip_engine.score_host(np.zeros((2, 8, 8, 4), np.uint8))
    with pytest.raises(ValueError):
        hip_engine.score_device(0, 2, 8, 8)
    with pytest.raises(ValueError):
        hip_engine.score_host(np.zeros((2, 8, 8, 3), np.uint8), flags=0)


def test_full_size_properties_1080p(hip_engine):
    """BASELINE-size frames: size-independent properties + an oracle check on a sample."""
    import torch

    n, h, w = 96, 1080, 1920
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    x[40:48] = x[40]                              # a run of identical frames -> zero SAD
    torch.cuda.synchronize()
    whole = hip_engine.score_device(x.data_ptr(), n, h, w, flags=NOEDGE)
    # histogram totals and byte sums against an independent reduction
    assert (whole["hist"].sum(axis=1) == h * w).all()
    sums = x.view(n, -1).to(torch.int64).sum(dim=1).cpu().numpy()
    assert np.array_equal(whole["byte_sum"], sums.astype(np.uint64))
    assert not whole["sad_h"][0] and (whole["sad_v"][41:48] == 0).all() and whole["sad_v"][40] > 0
    # chunking invariance ("linearity" in time): two halves with a halo == the whole batch
    k = 37
    a = hip_engine.score_device(x.data_ptr(), k, h, w, flags=NOEDGE)
    b = hip_engine.score_device(x[k:].data_ptr(), n - k, h, w, d_prev=x[k - 1].data_ptr(), flags=NOEDGE)
    same(np.concatenate([a, b]), whole)
    # idempotence
    same(hip_engine.score_device(x.data_ptr(), n, h, w, flags=NOEDGE), whole)
    # symmetry of |a-b|: reversing time moves each SAD to its neighbour
    xr = torch.flip(x, dims=[0]).contiguous()
    torch.cuda.synchronize()                       # the engine launches on its own stream
    rev = hip_engine.score_device(xr.data_ptr(), n, h, w, flags=NOEDGE)
    assert np.array_equal(rev["sad_s"][1:], whole["sad_s"][1:][::-1])
    # oracle on a sample
    sample = x[:3].cpu().numpy()
    same(whole[:3], orc.score_batch(sample))


@pytest.mark.parametrize("clip", ["scenes_a", "fades_b", "ragged_c", "uniform_u"])
def test_reference_golden_runs_through_hip(golden, hip_engine, clip):
    frames = golden_clip(golden, clip)
    for name in golden["clips"][clip]["results"]:
        cls_name, kwargs, with_stats = golden["configs"][name]
        got = run_config(frames, cls_name, kwargs, with_stats, hip_engine)      # (incl. the edge term: PSD_ERR_UNSUPPORTED fails the test)
        assert_same_run(got, golden["clips"][clip]["results"][name], f"{clip}/{name}")


def test_per_frame_api_on_gpu(golden, hip_engine):
    frames = golden_clip(golden, "fades_b")
    det = psd.ContentDetector(engine=hip_engine)
    cuts = []
    for i, f in enumerate(frames):
        cuts += det.process_frame(psd.FrameTimecode(i, 25.0), f)
    assert [c.frame_num for c in cuts] == golden["clips"]["fades_b"]["results"]["content_default"]["cuts"]
    hd = psd.HistogramDetector.calculate_histogram(frames[0], bins=128, engine=hip_engine)
    assert hd.shape == (128,) and abs(float(np.sqrt((hd.astype(np.float64) ** 2).sum())) - 1.0) < 1e-6


def test_edges_match_oracle(hip_engine):
    from pyscenedetect_amd.synth import make_clip

    frames, _ = make_clip(9, 12, 90, 160, shot_len=(4, 6))
    got = hip_engine.score_host(frames, flags=E.SCORE_ALL)
    want = oracle_score(frames, edges=True)
    same(got, want, FIELDS + ("edge_xor",))


def test_edge_maps_match_oracle(hip_engine):
    """Dilated Canny edge maps, pixel for pixel (several sizes, kernels, flat and noisy content)."""
    from oracle.detectors_np import edge_map
    from pyscenedetect_amd.synth import make_clip

    cases = []
    frames, _ = make_clip(17, 3, 90, 160, shot_len=(1, 1))
    cases += [(f, 0) for f in frames] + [(frames[0], 3), (frames[1], 9)]
    frames, _ = make_clip(18, 2, 131, 197, shot_len=(1, 1), noise=6.0)   # odd size, > 1 hysteresis tile
    cases += [(f, 0) for f in frames]
    rng = np.random.default_rng(5)
    cases.append((rng.integers(0, 256, (70, 100, 3), dtype=np.uint8), 5))
    # widths that are multiples of the 128-px tile take sobel_nms_tile_kernel: one, two and three tiles across, partial
    # bottom tiles, noise (every pixel a candidate) and smoothed noise (sparse candidates next to the replicated borders)
    cases.append((rng.integers(0, 256, (37, 128, 3), dtype=np.uint8), 3))
    cases.append((rng.integers(0, 256, (100, 256, 3), dtype=np.uint8), 5))
    coarse = rng.integers(0, 256, (9, 48, 3), dtype=np.uint8)
    cases.append((np.repeat(np.repeat(coarse, 8, axis=0), 8, axis=1)[:67], 7))     # 67 x 384, blocky
    cases.append((np.full((40, 50, 3), 77, np.uint8), 0))
    cases.append((np.zeros((9, 9, 3), np.uint8), 3))
    for frame, k in cases:
        h, w, _ = frame.shape
        buf = hip_engine.alloc(frame.nbytes)
        buf.upload(frame.reshape(-1))
        got = hip_engine.edge_map(buf.ptr, h, w, edge_kernel=k)
        want = edge_map(frame, k)
        assert np.array_equal(got, want), f"edge map differs for {h}x{w} k={k}: {np.count_nonzero(got != want)} px"
        buf.free()


def test_edges_with_prev_and_chunks(hip_engine):
    from pyscenedetect_amd.synth import make_clip

    frames, _ = make_clip(23, 20, 72, 128, shot_len=(4, 7))
    want = oracle_score(frames, edges=True)
    got = hip_engine.score_host(frames, flags=E.SCORE_ALL)
    same(got, want, FIELDS + ("edge_xor",))
    # split with a halo: identical
    a = hip_engine.score_host(frames[:9], flags=E.SCORE_EDGES)
    b = hip_engine.score_host(frames[9:], prev=frames[8], flags=E.SCORE_EDGES)
    assert np.array_equal(np.concatenate([a, b])["edge_xor"], want["edge_xor"])
    with pytest.raises(ValueError):
        hip_engine.score_host(frames[:2], flags=E.SCORE_EDGES, edge_kernel=4)


@pytest.mark.parametrize("shape", [((180, 320), (144, 256)), ((200, 300), (100, 150)), ((97, 131), (41, 77)),
                                   ((1080, 1920), (144, 256)), ((64, 64), (64, 64))])
def test_resize_matches_oracle(hip_engine, shape):
    import cv2  # the oracle shim

    (sh, sw), (dh, dw) = shape
    rng = np.random.default_rng(sh + dw)
    src = rng.integers(0, 256, (3, sh, sw, 3), dtype=np.uint8)
    a = hip_engine.alloc(src.nbytes)
    a.upload(src.reshape(-1))
    b = hip_engine.alloc(3 * dh * dw * 3)
    hip_engine.resize_device(a.ptr, 3, sh, sw, b.ptr, dh, dw)
    got = b.download().reshape(3, dh, dw, 3)
    for i in range(3):
        want = cv2.resize(src[i], (dw, dh))
        assert np.array_equal(got[i], want), f"resize {sh}x{sw}->{dh}x{dw}: {np.count_nonzero(got[i] != want)} bytes differ"


def test_auto_downscale_golden_through_hip(golden, hip_engine):
    """Reference default pipeline: frames wider than 256 px are downscaled before scoring."""
    frames = golden_clip(golden, "wide_d")
    for name in golden["clips"]["wide_d"]["results"]:
        cls_name, kwargs, with_stats = golden["configs"][name]
        got = run_config(frames, cls_name, kwargs, with_stats, hip_engine, auto_downscale=True)
        assert_same_run(got, golden["clips"]["wide_d"]["results"][name], f"wide_d/{name}")


def test_4k_batch_properties_and_oracle_sample(hip_engine):
    """BASELINE config 3 size (3840x2160): fused luma pass + HSV pass, checked by properties and a sample."""
    import torch

    n, h, w = 24, 2160, 3840
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda", generator=g)
    x[5] = 255                                   # byte_sum = 6.3e9 > 2^32: the u64 path
    x[6] = 0
    x[7] = x[6]
    torch.cuda.synchronize()
    r = hip_engine.score_device(x.data_ptr(), n, h, w, flags=NOEDGE)
    assert (r["hist"].sum(axis=1) == h * w).all()
    assert r["byte_sum"][5] == 255 * 3 * h * w and r["byte_sum"][6] == 0 and r["hist"][5][255] == h * w
    assert r["sad_v"][6] == 255 * h * w and r["sad_s"][6] == 0 and r["sad_h"][7] == 0 and r["sad_v"][7] == 0
    sums = x.view(n, -1).to(torch.int64).sum(dim=1).cpu().numpy()
    assert np.array_equal(r["byte_sum"], sums.astype(np.uint64))
    same(r[:2], orc.score_batch(x[:2].cpu().numpy()))
    luma = hip_engine.score_device(x.data_ptr(), n, h, w, flags=E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM)
    same(luma, r, ("hist", "byte_sum"))


def test_shot_like_1080p_content_matches_oracle_and_finds_the_cuts(hip_engine):
    """Realistic content at full size: smooth shots + noise with hard cuts every 12 frames."""
    import torch

    from pyscenedetect_amd import epilogue

    n, h, w, shot = 48, 1080, 1920, 12
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    x = torch.empty((n, h, w, 3), dtype=torch.uint8, device="cuda")
    for s0 in range(0, n, shot):
        grid = torch.rand((1, 3, 9, 16), device="cuda", generator=g) * 255.0
        base = torch.nn.functional.interpolate(grid, size=(h, w), mode="bilinear", align_corners=True)[0].permute(1, 2, 0)
        for i in range(s0, s0 + shot):
            x[i] = (base + torch.randn((h, w, 3), device="cuda", generator=g) * 2.0).round().clamp(0, 255).to(torch.uint8)
    torch.cuda.synchronize()
    r = hip_engine.score_device(x.data_ptr(), n, h, w, flags=NOEDGE)
    same(r[10:14], orc.score_batch(x[10:14].cpu().numpy(), x[9].cpu().numpy()))
    sc = epilogue.content_scores(r, h, w)
    assert epilogue.content_cuts(sc["content_val"], 25.0, min_scene_len=5) == [12, 24, 36]


def test_randomised_shapes_flags_and_halos(hip_engine):
    """Seeded sweep: 48 random (n, H, W), with/without a preceding frame, random flag sets incl. edges."""
    rng = np.random.default_rng(2025)
    for case in range(48):
        n = int(rng.integers(1, 14))
        h = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 90, 131]))
        w = int(rng.choice([1, 2, 5, 16, 31, 64, 97, 160, 257]))
        flags = int(rng.integers(1, 16))
        kernel = int(rng.choice([0, 3, 5, 7])) if flags & E.SCORE_EDGES else 0
        style = case % 3
        if style == 0:
            fr = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        elif style == 1:   # smooth + noise (long weak edge chains)
            yy, xx = np.mgrid[0:h, 0:w]
            base = 128 + 90 * np.sin(xx / 6.0 + case) * np.cos(yy / 5.0)
            fr = np.clip(base[None, :, :, None] + rng.normal(0, 5, (n, h, w, 3)), 0, 255).astype(np.uint8)
        else:              # few distinct values (ties in max/min, flat regions)
            fr = (rng.integers(0, 3, (n, h, w, 3)) * 127).astype(np.uint8)
        pv = rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if case % 2 else None
        want = oracle_score(fr, pv, edges=bool(flags & E.SCORE_EDGES), kernel_size=kernel)
        got = hip_engine.score_host(fr, prev=pv, flags=flags, edge_kernel=kernel)
        tag = f"case {case}: n={n} {h}x{w} flags={flags} k={kernel} prev={pv is not None}"
        if flags & E.SCORE_HSV_SAD:
            for f in ("sad_h", "sad_s", "sad_v"):
                assert np.array_equal(got[f], want[f]), f"{tag}: {f}"
        if flags & (E.SCORE_LUMA_HIST | E.SCORE_BYTE_SUM):
            assert np.array_equal(got["hist"], want["hist"]) and np.array_equal(got["byte_sum"], want["byte_sum"]), tag
        if flags & E.SCORE_EDGES:
            assert np.array_equal(got["edge_xor"], want["edge_xor"]), f"{tag}: edge_xor {got['edge_xor']} vs {want['edge_xor']}"


def test_8k_frames(hip_engine):
    """7680x4320: sums exceed 32 bits (u64 records), 2 M groups per frame."""
    rng = np.random.default_rng(8)
    fr = rng.integers(0, 256, (3, 4320, 7680, 3), dtype=np.uint8)
    fr[2] = 255 - fr[1]
    same(hip_engine.score_host(fr, flags=NOEDGE), orc.score_batch(fr))


def test_pipelined_submissions_and_limits(hip_engine):
    rng = np.random.default_rng(4)
    clips = [rng.integers(0, 256, (5 + i, 40, 64, 3), dtype=np.uint8) for i in range(4)]
    bufs = []
    for c in clips:
        b = hip_engine.alloc(c.nbytes)
        b.upload(c.reshape(-1))
        bufs.append(b)
    for c, b in zip(clips, bufs):
        hip_engine.submit_device(b.ptr, len(c), 40, 64, flags=NOEDGE)
    with pytest.raises(ValueError, match="in flight"):
        hip_engine.submit_device(bufs[0].ptr, len(clips[0]), 40, 64, flags=NOEDGE)     # PSD_MAX_INFLIGHT = 4
    with pytest.raises(ValueError, match="expected n"):
        hip_engine.collect(99)                                                          # a bad collect consumes nothing
    for c in clips:                                                                     # results come back in order
        same(hip_engine.collect(len(c)), orc.score_batch(c))
    with pytest.raises(ValueError, match="nothing submitted"):
        hip_engine.collect(1)
    same(hip_engine.score_device(bufs[0].ptr, len(clips[0]), 40, 64, flags=NOEDGE), orc.score_batch(clips[0]))


def test_two_engines_in_two_threads():
    """Distinct engines may be driven from distinct threads concurrently (include/psd_engine.h)."""
    import threading

    rng = np.random.default_rng(6)
    clips = [rng.integers(0, 256, (40, 90, 160, 3), dtype=np.uint8) for _ in range(2)]
    want = [orc.score_batch(c) for c in clips]
    errors = []

    def work(i):
        try:
            with E.ScoringEngine(0) as eng:
                for _ in range(5):
                    got = eng.score_host(clips[i], flags=NOEDGE)
                    for f in FIELDS:
                        assert np.array_equal(got[f], want[i][f]), f
        except Exception as ex:  # noqa: BLE001
            errors.append(ex)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


# ---- HashDetector thumbnails (psd_hash_thumbs*) ---------------------------------------------------

HASH_CASES = [
    # (n, h, w, size)            what it exercises
    (3, 72, 128, 16),          # fractional x integer scales -> float run tables, 16-byte fast loads
    (3, 54, 96, 32),           # fractional both ways
    (4, 37, 53, 16),           # ragged width: byte-load path
    (2, 64, 64, 32),           # exact 2x2 -> rounding shift
    (2, 96, 128, 32),          # integer 3x4 box -> sum * (1.f/area)
    (2, 144, 256, 16),         # integer 9x16
    (1, 33, 47, 33),           # 1x identity vertically, fractional horizontally
    (2, 1080, 1920, 16),       # HashDetector default at 1080p (67.5 x 120)
    (1, 1080, 1920, 32),       # 33.75 x 60
    (1, 2160, 3840, 16),
]


@pytest.mark.parametrize("case", HASH_CASES)
def test_hash_thumbs_match_oracle(hip_engine, case):
    n, h, w, size = case
    rng = np.random.default_rng(h * 7 + w + size)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[0, : h // 2] = 255  # saturated half: rounding at the top of the range
    want = orc.hash_thumbs(frames, size)
    got_host = hip_engine.hash_thumbs_host(frames, size)
    assert np.array_equal(got_host, want), np.argwhere(got_host != want)[:4].tolist()
    buf = hip_engine.alloc(frames.nbytes)
    buf.upload(frames.reshape(-1))
    got_dev = hip_engine.hash_thumbs_device(buf.ptr, n, h, w, size)
    assert np.array_equal(got_dev, want)
    # hash bits: native epilogue == oracle's numpy/C back half
    from pyscenedetect_amd import epilogue

    for hs in (8, size // 2):
        assert np.array_equal(epilogue.hash_bits(got_dev, hs), orc.hash_bits(want, hs))


def test_hash_thumbs_padded_rows_and_smooth_content(hip_engine):
    """Strided host frames (decoder padding) and low-contrast content, where float rounding order shows."""
    rng = np.random.default_rng(99)
    n, h, w = 5, 90, 160
    base = rng.integers(100, 110, (n, h, w + 6, 3), dtype=np.uint8)
    frames = base[:, :, 3 : 3 + w]
    got = hip_engine.hash_thumbs_host(frames, 16)
    assert np.array_equal(got, orc.hash_thumbs(np.ascontiguousarray(frames), 16))


@pytest.mark.parametrize("case", [(3, 24, 520, 32), (3, 24, 80, 32), (2, 54, 32, 16 + 32), (4, 20, 20, 32), (2, 37, 53, 64), (1, 20, 40, 32),
                                  (2, 16, 16, 16 + 1)])
def test_hash_thumbs_of_frames_smaller_than_the_thumbnail(hip_engine, case):
    """``cv2.resize(INTER_AREA)`` that does not shrink along both axes is OpenCV's bilinear emulation with area-mode coefficients:
    the reference accepts such frames (a 24-row frame, 32 x 32 thumbnails), rounds 1-4 refused them (round 5,
    tools/fuzz_host_vs_reference.py)."""
    n, h, w, size = case
    rng = np.random.default_rng(h * 11 + w + size)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    frames[0, : h // 2] = 255
    want = orc.hash_thumbs(frames, size)
    assert np.array_equal(hip_engine.hash_thumbs_host(frames, size), want)
    pad = np.zeros((n, h, w + 5, 3), np.uint8)            # padded rows (a decoder's stride)
    pad[:, :, 2:2 + w] = frames
    assert np.array_equal(hip_engine.hash_thumbs_host(pad[:, :, 2:2 + w], size), want)
    import cv2  # the oracle's shim: the reference's own two calls

    for t in range(n):
        assert np.array_equal(want[t], cv2.resize(cv2.cvtColor(frames[t], cv2.COLOR_BGR2GRAY), (size, size), interpolation=cv2.INTER_AREA))


def test_hash_thumbs_invalid_arguments(hip_engine):
    frames = np.zeros((1, 20, 40, 3), np.uint8)
    with pytest.raises(ValueError):
        hip_engine.hash_thumbs_host(frames, 0)
    with pytest.raises(ValueError):
        hip_engine.hash_thumbs_host(np.zeros((1, 20, 40), np.uint8), 8)
    assert hip_engine.hash_thumbs_host(np.zeros((0, 20, 40, 3), np.uint8), 8).shape == (0, 8, 8)


def test_hash_detector_per_frame_api_and_mixed_pass(golden, hip_engine):
    frames = golden_clip(golden, "fades_b")
    det = psd.HashDetector(engine=hip_engine)
    cuts = []
    for i, f in enumerate(frames):
        cuts += det.process_frame(psd.FrameTimecode(i, 25.0), f)
    assert [c.frame_num for c in cuts] == golden["clips"]["fades_b"]["results"]["hash_default"]["cuts"]
    # hash + content in one SceneManager pass: one upload, two device products
    sm = psd.SceneManager(engine=hip_engine)
    sm.auto_downscale = False
    sm.add_detector(psd.HashDetector(engine=hip_engine))
    sm.add_detector(psd.ContentDetector(engine=hip_engine))
    sm.detect_scenes(psd.ArrayVideoStream(frames, 25.0))
    res = golden["clips"]["fades_b"]["results"]
    assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == sorted(set(res["hash_default"]["cuts"]) | set(res["content_default"]["cuts"]))


def test_hash_thumbs_batch_properties_1080p(hip_engine):
    """Full-size batch: thumbnails of a batch equal the thumbnails of its frames taken one at a time, a frame
    and its copy hash identically, and a constant frame gives a constant thumbnail of its grey level."""
    import torch

    n, h, w = 64, 1080, 1920
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
    x[5] = x[4]
    x[7] = 200
    torch.cuda.synchronize()
    all_t = hip_engine.hash_thumbs_device(x.data_ptr(), n, h, w, 16)
    for i in (0, 4, 63):
        one = hip_engine.hash_thumbs_device(x[i].data_ptr(), 1, h, w, 16)
        assert np.array_equal(one[0], all_t[i])
    assert np.array_equal(all_t[4], all_t[5])
    assert (all_t[7] == 200).all()
    sample = x[:2].cpu().numpy()
    assert np.array_equal(all_t[:2], orc.hash_thumbs(sample, 16))


def test_hash_thumbs_randomised_shapes(hip_engine):
    """Random frame sizes, thumbnail sizes and strides: every kernel variant (LDS-DMA stream, register loads,
    byte loads; float run tables and integer boxes) against the oracle."""
    rng = np.random.default_rng(20250921)
    for _ in range(40):
        size = int(rng.choice([4, 7, 8, 16, 24, 32, 64]))
        h = int(rng.integers(size, 6 * size + 40))
        w = int(rng.integers(size, 8 * size + 60))
        if rng.random() < 0.5:
            w = max(size, (w + 15) // 16 * 16)          # fast paths need width % 16 == 0
        if rng.random() < 0.3:
            h, w = size * int(rng.integers(1, 5)), max(size, size * int(rng.integers(1, 6)))  # integer boxes
        n = int(rng.integers(1, 5))
        pad = int(rng.choice([0, 0, 16, 5]))
        base = rng.integers(0, 256, (n, h, w + pad, 3), dtype=np.uint8)
        frames = base[:, :, : w]
        want = orc.hash_thumbs(np.ascontiguousarray(frames), size)
        got = hip_engine.hash_thumbs_host(frames, size)
        assert np.array_equal(got, want), (n, h, w, size, pad, np.argwhere(got != want)[:3].tolist())
        # the same frames resident in HBM with their padding: strided rows take the register / byte-load kernels
        buf = hip_engine.alloc(base.nbytes)
        buf.upload(base.reshape(-1))
        got_dev = hip_engine.hash_thumbs_device(buf.ptr, n, h, w, size, row_stride=(w + pad) * 3,
                                                frame_stride=h * (w + pad) * 3)
        assert np.array_equal(got_dev, want), ("device", n, h, w, size, pad)
        buf.free()


# ---- SceneManager.interpolation = NEAREST / AREA --------------------------------------------------------------

@pytest.mark.parametrize("shape", [((180, 320), (144, 256)), ((200, 300), (100, 150)), ((97, 131), (41, 77)),
                                   ((64, 96), (32, 32)), ((1080, 1920), (144, 256)), ((90, 120), (30, 40)),
                                   ((50, 70), (50, 70))])
@pytest.mark.parametrize("mode", ["NEAREST", "AREA", "LANCZOS4", "CUBIC"])
def test_resize_nearest_and_area_match_oracle(hip_engine, shape, mode):
    import cv2  # the oracle shim

    (sh, sw), (dh, dw) = shape
    inter = {"NEAREST": cv2.INTER_NEAREST, "AREA": cv2.INTER_AREA, "LANCZOS4": cv2.INTER_LANCZOS4, "CUBIC": cv2.INTER_CUBIC}[mode]
    rng = np.random.default_rng(sh * 3 + dw)
    src = rng.integers(0, 256, (2, sh, sw, 3), dtype=np.uint8)
    src[1, : sh // 2] = 255
    a = hip_engine.alloc(src.nbytes)
    a.upload(src.reshape(-1))
    b = hip_engine.alloc(2 * dh * dw * 3)
    hip_engine.resize_device(a.ptr, 2, sh, sw, b.ptr, dh, dw, interpolation=inter)
    got = b.download().reshape(2, dh, dw, 3)
    for i in range(2):
        want = cv2.resize(src[i], (dw, dh), interpolation=inter)
        assert np.array_equal(got[i], want), f"{mode} {sh}x{sw}->{dh}x{dw}: {np.count_nonzero(got[i] != want)} bytes differ"


def test_resize_unsupported_modes(hip_engine):
    a = hip_engine.alloc(64 * 64 * 3)
    b = hip_engine.alloc(128 * 128 * 3)
    for interp in (5, -1, 7):                                                          # (no such filter: cv2's 0 .. 4 are all there)
        with pytest.raises(NotImplementedError):
            hip_engine.resize_device(a.ptr, 1, 64, 64, b.ptr, 32, 32, interpolation=interp)


@pytest.mark.parametrize("shape", [((64, 64), (128, 128)), ((45, 80), (90, 160)), ((37, 53), (55, 80)), ((72, 128), (36, 256)), ((9, 5), (3, 2)),
                                   ((4, 4), (31, 17)), ((2160, 3840), (144, 256))])
@pytest.mark.parametrize("mode", ["LANCZOS4", "CUBIC"])
def test_resize_lanczos4_enlarging_and_extreme_shapes(hip_engine, shape, mode):
    """cv2.resize(INTER_LANCZOS4 / INTER_CUBIC) where most taps fall outside the image (tiny sources: every tap clamped), enlargements, mixed
    directions and the 15-fold reduction of a 4K frame -- OpenCV's integer arithmetic, byte for byte."""
    import cv2  # the oracle shim

    inter = {"LANCZOS4": cv2.INTER_LANCZOS4, "CUBIC": cv2.INTER_CUBIC}[mode]       # (CUBIC: the default form, PSD_CUBIC_FORM unset)

    (sh, sw), (dh, dw) = shape
    rng = np.random.default_rng(sh * 7 + dw)
    src = rng.integers(0, 256, (2, sh, sw, 3), dtype=np.uint8)
    src[1, :, : max(1, sw // 2)] = (255, 0, 255)               # (hard edges: the filter's negative lobes saturate at both ends)
    a = hip_engine.alloc(src.nbytes)
    a.upload(src.reshape(-1))
    b = hip_engine.alloc(2 * dh * dw * 3)
    hip_engine.resize_device(a.ptr, 2, sh, sw, b.ptr, dh, dw, interpolation=inter)
    got = b.download().reshape(2, dh, dw, 3)
    for i in range(2):
        want = cv2.resize(src[i], (dw, dh), interpolation=inter)
        assert np.array_equal(got[i], want), f"{mode} {sh}x{sw}->{dh}x{dw}: {np.count_nonzero(got[i] != want)} bytes differ"
    if mode == "LANCZOS4" or dh * dw > 100:
        assert got.min() == 0 and got.max() == 255


@pytest.mark.parametrize("shape", [((64, 64), (128, 128)), ((45, 80), (90, 160)), ((37, 53), (55, 80)), ((72, 128), (36, 256)),
                                   ((72, 128), (100, 64)), ((90, 160), (91, 161)), ((16, 48), (48, 48))])
def test_resize_area_upscaling_matches_oracle(hip_engine, shape):
    """cv2.resize(INTER_AREA) that does not shrink along both axes (enlargements, and wider-but-lower / higher-but-narrower
    targets): OpenCV's bilinear emulation with area-mode coefficients; 16-byte aligned source rows take the staged kernel,
    the others the generic one."""
    import cv2  # the oracle shim

    (sh, sw), (dh, dw) = shape
    rng = np.random.default_rng(sh * 7 + dw)
    src = rng.integers(0, 256, (3, sh, sw, 3), dtype=np.uint8)
    a = hip_engine.alloc(src.nbytes)
    a.upload(src.reshape(-1))
    b = hip_engine.alloc(3 * dh * dw * 3)
    hip_engine.resize_device(a.ptr, 3, sh, sw, b.ptr, dh, dw, interpolation=cv2.INTER_AREA)
    got = b.download().reshape(3, dh, dw, 3)
    for i in range(3):
        want = cv2.resize(src[i], (dw, dh), interpolation=cv2.INTER_AREA)
        assert np.array_equal(got[i], want), f"AREA {sh}x{sw}->{dh}x{dw}: {np.count_nonzero(got[i] != want)} bytes differ"
    a.free()
    b.free()


def test_downscale_interpolation_modes_golden_through_hip(golden, hip_engine):
    frames = golden_clip(golden, "wide_d")
    for mode, runs in golden["interp"].items():
        for name, want in runs.items():
            cls_name, kwargs, with_stats = golden["configs"][name]
            got = run_config(frames, cls_name, kwargs, with_stats, hip_engine, auto_downscale=True, interpolation=mode)
            assert_same_run(got, want, f"wide_d/{mode}/{name}")


def test_hash_thumbs_host_batches_larger_than_one_staging_chunk(hip_engine):
    """Host frames are staged in 256 MiB chunks; the thumbnails must not depend on where the chunks end."""
    rng = np.random.default_rng(4)
    n, h, w = 47, 1080, 1920                      # 292 MB: two chunks
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    got = hip_engine.hash_thumbs_host(frames, 16)
    one_by_one = np.stack([hip_engine.hash_thumbs_host(frames[i : i + 1], 16)[0] for i in (0, 42, 43, 46)])
    assert np.array_equal(got[[0, 42, 43, 46]], one_by_one)
    assert np.array_equal(got[-1], orc.hash_thumbs(frames[-1:], 16)[0])


@pytest.mark.parametrize("size,hash_size", [(32, 16), (16, 8), (64, 16), (12, 3), (48, 11), (8, 8)])
def test_hash_bits_on_the_device_equal_the_host_epilogue(hip_engine, size, hash_size):
    """``psd_hash_bits_device`` (round 6: HashDetector's scaling, float64 DCT, float32 median and threshold on the device, beside the
    thumbnail kernel) == ``psd_epilogue_hash_bits`` of the thumbnails == the oracle, bit for bit: shot-like frames, uniform noise,
    constant frames (every AC coefficient rounding noise: the bits only agree if every sum runs in the same order), all-black frames
    (the maximum replaced by 1, hash_detector.py:132-135), odd hash sizes (an odd count's median is one element) and transforms of
    12 ... 64 points."""
    from oracle import lib as orc
    from pyscenedetect_amd import epilogue
    from pyscenedetect_amd.synth import make_clip

    rng = np.random.default_rng(size * 100 + hash_size)
    clip, _ = make_clip(size + hash_size, 24, 90, 160, shot_len=(5, 9))
    frames = np.concatenate([clip, rng.integers(0, 256, (8, 90, 160, 3), dtype=np.uint8),
                             np.full((3, 90, 160, 3), 137, np.uint8), np.zeros((2, 90, 160, 3), np.uint8),
                             np.clip(rng.integers(-1, 2, (6, 90, 160, 3)) + 60, 0, 255).astype(np.uint8)])
    n = len(frames)
    buf = device_copy(hip_engine, frames)
    bits, thumbs = hip_engine.hash_bits_device(buf.ptr, n, 90, 160, size, hash_size, want_thumbs=True)
    assert np.array_equal(thumbs, hip_engine.hash_thumbs_device(buf.ptr, n, 90, 160, size))
    assert np.array_equal(thumbs, orc.hash_thumbs(frames, size))
    host = epilogue.hash_bits(thumbs, hash_size)
    assert bits.shape == (n, hash_size * hash_size) and np.array_equal(bits, host.reshape(n, -1))
    assert np.array_equal(bits, orc.hash_bits(thumbs, hash_size).reshape(n, -1).astype(np.uint8))
    assert np.array_equal(hip_engine.hash_bits_device(buf.ptr, n, 90, 160, size, hash_size), bits)       # bits alone
    buf.free()


def test_hash_bits_on_the_device_refuse_what_does_not_fit(hip_engine):
    frames = np.zeros((2, 300, 300, 3), np.uint8)
    buf = device_copy(hip_engine, frames)
    with pytest.raises(NotImplementedError):
        hip_engine.hash_bits_device(buf.ptr, 2, 300, 300, 128, 64)          # a 128-point transform: the two-step form takes it
    with pytest.raises(ValueError):
        hip_engine.hash_bits_device(buf.ptr, 2, 300, 300, 16, 17)           # more frequencies than points
    assert hip_engine.hash_bits_device(buf.ptr, 2, 300, 300, 16, 8).shape == (2, 64)
    buf.free()
