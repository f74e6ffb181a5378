"""CPU, world_size 2 over gloo: sharding by frame range (1-frame halo) and by clip, with the
all-gather of score records, gives the same records and cut lists as a single process."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from pyscenedetect_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 4097):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_assign_clips_balances():
    plan = D.assign_clips([100, 90, 50, 40, 10, 10], 2)
    assert sorted(sum(plan, [])) == list(range(6))
    loads = [sum([100, 90, 50, 40, 10, 10][i] for i in p) for p in plan]
    assert abs(loads[0] - loads[1]) <= 20


CORPUS = [(70, 36, 64), (50, 54, 96), (33, 24, 40), (90, 36, 64), (41, 48, 80)]
ALL_FOUR = {"content": {"min_scene_len": 8}, "adaptive": {"min_scene_len": 8}, "hist": {}, "threshold": {"threshold": 40}}


NO_HIST = {"content": {"min_scene_len": 8}, "adaptive": {"min_scene_len": 8}, "threshold": {"threshold": 40}}
# clips wider than 256 pixels: the reference's default pipeline (auto_downscale) resizes them, each resolution by its own factor
CORPUS_WIDE = [(40, 90, 320), (35, 135, 480), (25, 36, 64), (30, 180, 288), (22, 90, 320)]


class PackingOracle:
    """The oracle behind the interface of ScoringEngine.score_clips (sums_only: records without the histogram)."""

    def __init__(self):
        from oracle.detectors_np import OracleEngine

        self.eng = OracleEngine()
        self.asked_for_sums = []

    def score_clips(self, clips, flags, edge_kernel=0, sums_only=False, downscale=None, interpolation=1):
        from pyscenedetect_amd.engine import _sums_of, downscale_size

        self.asked_for_sums.append(sums_only)
        recs = [self.eng.score_host(c, flags=flags, edge_kernel=edge_kernel, downscale=downscale_size(c.shape[1], c.shape[2], downscale)[0],
                                    interpolation=interpolation) for c in clips]
        return [_sums_of(r) for r in recs] if sums_only else recs


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.detectors_np import OracleEngine
    from pyscenedetect_amd import epilogue
    from pyscenedetect_amd.synth import make_clip

    eng = OracleEngine()
    frames, _ = make_clip(31, 75, 36, 64, shot_len=(8, 16))
    recs = D.score_clip_sharded(eng, lambda a, b: frames[a:b], len(frames), flags=7)
    clips = [make_clip(40 + i, n, 24, 40, shot_len=(5, 9))[0] for i, n in enumerate((30, 11, 22, 5))]
    per_clip = D.score_clips_distributed(eng, clips, flags=7)
    # with a HistogramDetector's bin count the ranks exchange the sums and hist_diff (48 bytes per frame), not the histograms: the same
    # sums, and the values the host epilogue computes from the full records
    from pyscenedetect_amd._native import SUMS_DIFF_DTYPE
    slim = D.score_clips_distributed(eng, clips, flags=7, hist_diff_bins=100)
    for a, b in zip(slim, per_clip):
        assert a.dtype == SUMS_DIFF_DTYPE and len(a) == len(b) and all(np.array_equal(a[k], b[k]) for k in ("sad_h", "sad_s", "sad_v", "byte_sum"))
        want = epilogue.hist_cuts(b, 25.0, 0.2, 100, 15)[1]
        assert np.isnan(a["hist_diff"][0]) and np.array_equal(a["hist_diff"][1:].view(np.uint64), want[1:].view(np.uint64))
    sc = epilogue.content_scores(recs, 36, 64)
    cuts = epilogue.content_cuts(sc["content_val"], 25.0, min_scene_len=5)
    # mixed-resolution corpus, all four detectors (BASELINE.json config 5 in miniature)
    import json

    from pyscenedetect_amd.corpus import detect_corpus

    corpus = [make_clip(60 + i, n, h, w, shot_len=(16, 24))[0] for i, (n, h, w) in enumerate(CORPUS)]
    res = detect_corpus(eng, corpus, 25.0, ALL_FOUR)
    with open(os.path.join(tmp, f"corpus{rank}.json"), "w") as f:
        json.dump(res, f)
    # records without the histogram (ABI 3): ragged all-gather of 40-byte sums, one rank empty; and the clip flow over an
    # engine that packs clips and returns sums (what ScoringEngine.score_clips does when no detector reads the histogram)
    from pyscenedetect_amd._native import SUMS_DTYPE

    mine = np.zeros(0 if rank == 0 else 6 + rank, SUMS_DTYPE)
    mine["sad_v"] = np.arange(len(mine)) + 100 * rank
    parts = D.all_gather_records(mine)
    assert [len(x) for x in parts] == [0] + [6 + r for r in range(1, world)] and parts[1].dtype == SUMS_DTYPE
    assert all(parts[r]["sad_v"].tolist() == list(range(100 * r, 100 * r + 6 + r)) for r in range(1, world))
    res = detect_corpus(PackingOracle(), corpus, 25.0, NO_HIST)
    with open(os.path.join(tmp, f"nohist{rank}.json"), "w") as f:
        json.dump(res, f)
    res = detect_corpus(PackingOracle(), corpus[:1], 25.0, NO_HIST)     # fewer clips than ranks: rank 1 sends nothing
    with open(os.path.join(tmp, f"nohist_one{rank}.json"), "w") as f:
        json.dump(res, f)
    # the reference's default pipeline (auto_downscale, the default of detect_corpus) on clips that it resizes: sharded by clip,
    # every rank resizes and scores its own, the 40-byte sums are all-gathered, every rank decides with the RESIZED size
    wide = [make_clip(80 + i, n, h, w, shot_len=(9, 14), noise=12.0)[0] for i, (n, h, w) in enumerate(CORPUS_WIDE)]
    res = {"all_four": detect_corpus(eng, wide, 25.0, ALL_FOUR), "packed_sums": detect_corpus(PackingOracle(), wide, 25.0, NO_HIST),
           "manual_area": detect_corpus(eng, wide, 25.0, NO_HIST, auto_downscale=False, downscale=2, interpolation=3)}
    with open(os.path.join(tmp, f"wide{rank}.json"), "w") as f:
        json.dump(res, f)
    np.save(os.path.join(tmp, f"recs{rank}.npy"), recs)
    np.save(os.path.join(tmp, f"cuts{rank}.npy"), np.array(cuts))
    for i, r in enumerate(per_clip):
        np.save(os.path.join(tmp, f"clip{i}_{rank}.npy"), r)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one(tmp_path):
    from oracle.detectors_np import OracleEngine
    from pyscenedetect_amd import epilogue
    from pyscenedetect_amd.synth import make_clip

    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    eng = OracleEngine()
    frames, _ = make_clip(31, 75, 36, 64, shot_len=(8, 16))
    want = eng.score_host(frames, flags=7)
    sc = epilogue.content_scores(want, 36, 64)
    want_cuts = epilogue.content_cuts(sc["content_val"], 25.0, min_scene_len=5)
    assert want_cuts
    for r in range(world):
        got = np.load(tmp_path / f"recs{r}.npy")
        assert got.tobytes() == want.tobytes()
        assert np.load(tmp_path / f"cuts{r}.npy").tolist() == want_cuts
    clips = [make_clip(40 + i, n, 24, 40, shot_len=(5, 9))[0] for i, n in enumerate((30, 11, 22, 5))]
    for i, c in enumerate(clips):
        ref = eng.score_host(c, flags=7)
        for r in range(world):
            assert np.load(tmp_path / f"clip{i}_{r}.npy").tobytes() == ref.tobytes()
    # corpus: 2 ranks == 1 process == the per-frame detectors through SceneManager
    import json

    import pyscenedetect_amd as psd
    from pyscenedetect_amd.corpus import detect_corpus

    corpus = [make_clip(60 + i, n, h, w, shot_len=(16, 24))[0] for i, (n, h, w) in enumerate(CORPUS)]
    single = detect_corpus(eng, corpus, 25.0, ALL_FOUR)
    assert any(any(v for v in clip.values()) for clip in single)
    for r in range(world):
        assert json.load(open(tmp_path / f"corpus{r}.json")) == single
    packing = PackingOracle()
    nohist = detect_corpus(packing, corpus, 25.0, NO_HIST)
    assert packing.asked_for_sums == [True]
    assert nohist == [{k: v for k, v in clip.items() if k != "hist"} for clip in single]
    for r in range(world):
        assert json.load(open(tmp_path / f"nohist{r}.json")) == nohist
        assert json.load(open(tmp_path / f"nohist_one{r}.json")) == nohist[:1]
    for clip, res in zip(corpus, single):
        sm = psd.SceneManager(engine=eng)
        sm.auto_downscale = False
        sm.add_detector(psd.ContentDetector(min_scene_len=8, engine=eng))
        sm.detect_scenes(psd.ArrayVideoStream(clip, 25.0))
        assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == res["content"]
    _check_wide(tmp_path, world, eng)


def _check_wide(tmp_path, world, eng):
    """Every rank's result for the resized corpus == one process == a default SceneManager per clip (auto_downscale on)."""
    import json

    import pyscenedetect_amd as psd
    from pyscenedetect_amd.corpus import detect_corpus
    from pyscenedetect_amd.engine import downscale_size
    from pyscenedetect_amd.synth import make_clip

    wide = [make_clip(80 + i, n, h, w, shot_len=(9, 14), noise=12.0)[0] for i, (n, h, w) in enumerate(CORPUS_WIDE)]
    assert sum(downscale_size(h, w, "auto")[0] > 1.0 for _, h, w in CORPUS_WIDE) == 4
    single = {"all_four": detect_corpus(eng, wide, 25.0, ALL_FOUR)}
    single["packed_sums"] = [{k: v for k, v in clip.items() if k != "hist"} for clip in single["all_four"]]
    single["manual_area"] = detect_corpus(eng, wide, 25.0, NO_HIST, auto_downscale=False, downscale=2, interpolation=3)
    assert single["all_four"] != detect_corpus(eng, wide, 25.0, ALL_FOUR, auto_downscale=False), "the resize changes decisions on this corpus"
    for r in range(world):
        assert json.load(open(tmp_path / f"wide{r}.json")) == single
    for clip, res in zip(wide, single["all_four"]):
        sm = psd.SceneManager(engine=eng)                    # auto_downscale is the default, as in the reference
        sm.add_detector(psd.ContentDetector(min_scene_len=8, engine=eng))
        sm.add_detector(psd.HistogramDetector(engine=eng))
        sm.detect_scenes(psd.ArrayVideoStream(clip, 25.0))
        assert [c.frame_num for c in sm.get_cut_list(show_warning=False)] == sorted(set(res["content"]) | set(res["hist"]))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_ranks_equal_one(tmp_path, world):
    """The same flows over four and eight ranks -- the GPU counts of BASELINE's metric (1 vs 2 vs 4 vs 8 give
    byte-identical records and cut lists).  More ranks than some shards have clips: the one-clip corpus leaves all ranks
    but one without work; the frame-range shards of the 75-frame clip are 19 / 19 / 19 / 18 (10 / 10 / 10 / 9 ... ) frames."""
    import json

    from oracle.detectors_np import OracleEngine
    from pyscenedetect_amd.corpus import detect_corpus
    from pyscenedetect_amd.synth import make_clip

    port = 31500 + world * 16 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    eng = OracleEngine()
    frames, _ = make_clip(31, 75, 36, 64, shot_len=(8, 16))
    want = eng.score_host(frames, flags=7)
    corpus = [make_clip(60 + i, n, h, w, shot_len=(16, 24))[0] for i, (n, h, w) in enumerate(CORPUS)]
    single = detect_corpus(eng, corpus, 25.0, ALL_FOUR)
    nohist = [{k: v for k, v in clip.items() if k != "hist"} for clip in single]
    for r in range(world):
        assert np.load(tmp_path / f"recs{r}.npy").tobytes() == want.tobytes()
        assert json.load(open(tmp_path / f"corpus{r}.json")) == single
        assert json.load(open(tmp_path / f"nohist{r}.json")) == nohist
        assert json.load(open(tmp_path / f"nohist_one{r}.json")) == nohist[:1]
    _check_wide(tmp_path, world, eng)


# ---- the exchange step through the C-ABI (psd_comm_* / psd_allgather_host) instead of torch.distributed ---------------------------
class AbiEngine:
    """The oracle engine with a handle of the CPU build of the C-ABI (oracle/libpsd_oracle_abi.so): what NativeComm needs."""

    def __init__(self, lib):
        import ctypes

        from oracle.detectors_np import OracleEngine

        self._oracle = OracleEngine()
        h = ctypes.c_void_p()
        assert lib.psd_create(0, ctypes.byref(h)) == 0
        self._h = h

    def score_host(self, *a, **kw):
        return self._oracle.score_host(*a, **kw)


def _abi_lib():
    import ctypes
    import subprocess

    from pyscenedetect_amd import _native

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libpsd_oracle_abi.so"])
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libpsd_oracle_abi.so"))
    for name in ("psd_create", "psd_last_error", "psd_comm_unique_id", "psd_comm_create", "psd_comm_destroy", "psd_allgather_host",
                 "psd_allgather_scores"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _native.SYMBOLS[name]
    return lib


def _native_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import json

    from pyscenedetect_amd import corpus
    from pyscenedetect_amd._native import RECORD_DTYPE, SUMS_DTYPE
    from pyscenedetect_amd.synth import make_clip

    lib = _abi_lib() if rank == 0 else None
    dist.barrier()                                      # (one rank builds the library, the others load it)
    lib = lib or _abi_lib()
    eng = AbiEngine(lib)
    assert D.native_comm_for(eng, None) is None         # gloo and no stand-in handed in: the flow keeps to torch.distributed
    D._native_comms.clear()
    comm = D.native_comm_for(eng, None, lib=lib)
    assert comm is not None and D.native_comm_for(eng, None) is comm and comm.n_ranks == world
    # ragged blocks of both kinds, a rank with nothing
    for dtype in (SUMS_DTYPE, RECORD_DTYPE):
        counts = [0 if r == 1 else 3 + 2 * r for r in range(world)]
        mine = np.zeros(counts[rank], dtype)
        mine["sad_v"] = np.arange(counts[rank]) + 1000 * rank
        parts = comm.all_gather_host(mine, counts)
        assert [len(x) for x in parts] == counts and all(x.dtype == dtype for x in parts)
        assert all(parts[r]["sad_v"].tolist() == list(range(1000 * r, 1000 * r + counts[r])) for r in range(world))
    # a rank whose records do not match the counts: everybody completes the collective, that rank raises afterwards
    counts = [4] * world
    mine = np.zeros(3 if rank == world - 1 else 4, SUMS_DTYPE)
    try:
        comm.all_gather_host(mine, counts)
        assert rank != world - 1
    except ValueError as ex:
        assert rank == world - 1 and "contributes 3" in str(ex)
    # the sharded flows through it: default pipeline on the resized corpus, the small corpus, fewer clips than ranks
    before = comm.exchanges
    wide = [make_clip(80 + i, n, h, w, shot_len=(9, 14), noise=12.0)[0] for i, (n, h, w) in enumerate(CORPUS_WIDE)]
    small = [make_clip(60 + i, n, h, w, shot_len=(16, 24))[0] for i, (n, h, w) in enumerate(CORPUS)]
    res = {"wide_all_four": corpus.detect_corpus(eng, wide, 25.0, ALL_FOUR), "wide_sums": corpus.detect_corpus(eng, wide, 25.0, NO_HIST),
           "small": corpus.detect_corpus(eng, small, 25.0, ALL_FOUR), "one": corpus.detect_corpus(eng, small[:1], 25.0, NO_HIST)}
    assert comm.exchanges == before + 4                 # every flow took the native exchange, once
    with open(os.path.join(tmp, f"native{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_exchange_over_the_c_abi_stand_in(tmp_path, world):
    """``score_clips_distributed`` with the exchange the C-ABI advertises -- ``NativeComm`` over ``psd_comm_*`` / ``psd_allgather_host``,
    counts from the plan, one collective -- executed over the shared-memory stand-in of ``oracle/abi_cpu.c`` (the RCCL implementation
    needs GPUs): every rank's cut lists == one process."""
    import json

    from oracle.detectors_np import OracleEngine
    from pyscenedetect_amd import corpus
    from pyscenedetect_amd.synth import make_clip

    port = 33500 + world * 16 + (os.getpid() % 1500)
    mp.spawn(_native_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    eng = OracleEngine()
    wide = [make_clip(80 + i, n, h, w, shot_len=(9, 14), noise=12.0)[0] for i, (n, h, w) in enumerate(CORPUS_WIDE)]
    small = [make_clip(60 + i, n, h, w, shot_len=(16, 24))[0] for i, (n, h, w) in enumerate(CORPUS)]
    single = {"wide_all_four": corpus.detect_corpus(eng, wide, 25.0, ALL_FOUR), "wide_sums": corpus.detect_corpus(eng, wide, 25.0, NO_HIST),
              "small": corpus.detect_corpus(eng, small, 25.0, ALL_FOUR), "one": corpus.detect_corpus(eng, small[:1], 25.0, NO_HIST)}
    for r in range(world):
        assert json.load(open(tmp_path / f"native{r}.json")) == single
