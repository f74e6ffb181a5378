"""Whole-corpus detection: many clips (mixed resolutions), several detectors, one or many GPUs.

This is the batch face of the engine for BASELINE.json configs 4-5: clips are packed onto the
GPUs of a node (one process per GPU, sharded by clip), every clip is scored ONCE for the union of
what the requested detectors need, the per-frame records are all-gathered (RCCL), and every rank
runs the native decision epilogues -- so the result does not depend on the number of GPUs.

What is computed per clip is what the reference computes per video: its benchmark runs
``detect(video, detector_cls())`` (``benchmark/__main__.py:44-61``), i.e. a ``SceneManager`` with
``auto_downscale=True`` whose decode thread resizes every frame to about 256 pixels width
(``scene_manager.py:110,123-140,666-678``) before any detector sees it.  ``detect_corpus`` does the
same by default (``auto_downscale=True``): every resolution of the corpus gets its own factor and the
packed batches go through ``psd_score_segments_downscaled_device``.
"""

import numpy as np

from pyscenedetect_amd import _native, epilogue

#: detector name -> (score flags, default parameters); parameters mirror the reference constructors.
DETECTORS = {
    "content": (_native.SCORE_HSV_SAD, dict(threshold=27.0, min_scene_len=15, weights=(1.0, 1.0, 1.0, 0.0), filter_mode=0)),
    "adaptive": (_native.SCORE_HSV_SAD, dict(adaptive_threshold=3.0, min_scene_len=15, window_width=2, min_content_val=15.0,
                                            weights=(1.0, 1.0, 1.0, 0.0))),
    "hist": (_native.SCORE_LUMA_HIST, dict(threshold=0.20, bins=128, min_scene_len=15)),
    "threshold": (_native.SCORE_BYTE_SUM, dict(threshold=12, min_scene_len=15, fade_bias=0.0, add_final_scene=False, method=0)),
}


def required_flags(detectors: dict) -> int:
    flags = 0
    for name, params in detectors.items():
        base, defaults = DETECTORS[name]
        flags |= base
        w = (params or {}).get("weights", defaults.get("weights"))
        if w is not None and len(w) == 4 and w[3] > 0.0:
            flags |= _native.SCORE_EDGES
    return flags


def decide(records: np.ndarray, height: int, width: int, fps, detectors: dict) -> dict:
    """Cut lists of one clip from its records: ``{detector name: [frame numbers]}``."""
    out = {}
    scores = None
    for name, params in detectors.items():
        p = dict(DETECTORS[name][1])
        p.update(params or {})
        if name in ("content", "adaptive"):
            weights = p.pop("weights")
            scores = epilogue.content_scores(records, height, width, weights)
            if name == "content":
                out[name] = epilogue.content_cuts(scores["content_val"], fps, p["threshold"], p["min_scene_len"], p["filter_mode"])
            else:
                out[name] = epilogue.adaptive_cuts(scores["content_val"], fps, p["adaptive_threshold"], p["min_scene_len"],
                                                   p["window_width"], p["min_content_val"])[0]
        elif name == "hist":
            if records.dtype.names and "hist_diff" in records.dtype.names:
                # (the values came from the device, for THIS detector's bin count: score_clips(hist_diff_bins=) -- detect_corpus asks for them)
                out[name] = epilogue.hist_cuts_from_diff(records["hist_diff"], fps, p["threshold"], p["min_scene_len"])
            else:
                out[name] = epilogue.hist_cuts(records, fps, p["threshold"], p["bins"], p["min_scene_len"])[0]
        elif name == "threshold":
            out[name] = epilogue.threshold_cuts(records, height, width, fps, p["threshold"], p["min_scene_len"], p["fade_bias"],
                                                p["add_final_scene"], p["method"])[0]
        else:
            raise KeyError(name)
    return out


def scored_size(height: int, width: int, downscale=None) -> tuple[int, int]:
    """``(height, width)`` of the frames the detectors see for source frames of ``height x width``: what ``decide()`` divides
    by and what sizes the edge term's dilation kernel (``content_detector.py:36,39-46`` on the resized frame)."""
    from pyscenedetect_amd.engine import downscale_size

    _, dh, dw = downscale_size(height, width, downscale)
    return dh, dw


def score_clip(engine, clip, flags: int, edge_kernel: int = 0, downscale=None, interpolation: int = 1) -> np.ndarray:
    """Records of one clip held either in host memory (ndarray) or in HBM (anything with
    ``data_ptr()``/``shape``, e.g. a torch uint8 tensor on this engine's device); ``downscale`` as in ``score_clips``."""
    from pyscenedetect_amd.engine import downscale_size

    n, h, w, c = clip.shape
    if n == 0:
        return np.zeros(0, _native.RECORD_DTYPE)
    factor, dh, dw = downscale_size(h, w, downscale)
    if hasattr(clip, "data_ptr"):
        if c != 3 or not clip.is_contiguous():
            raise ValueError("device clips must be contiguous uint8[n,H,W,3]")
        if factor > 1.0:
            return engine.score_device_downscaled(clip.data_ptr(), n, h, w, dh, dw, flags=flags, edge_kernel=edge_kernel,
                                                  interpolation=interpolation)
        return engine.score_device(clip.data_ptr(), n, h, w, flags=flags, edge_kernel=edge_kernel)
    if factor > 1.0:
        return engine.score_host(clip[0:len(clip)], flags=flags, edge_kernel=edge_kernel, downscale=factor, interpolation=interpolation)
    return engine.score_host(clip[0:len(clip)], flags=flags, edge_kernel=edge_kernel)


def score_clips(engine, clips, flags: int, edge_kernel: int = 0, on_ready=None, sums_only: bool | None = None,
                downscale=None, interpolation: int = 1, hist_diff_bins: int | None = None) -> list[np.ndarray]:
    """Records of every clip; engines that can pack clips of one resolution into shared batches (``ScoringEngine.score_clips``:
    one launch per term per batch, SAD chain broken at clip starts) do so, others score clip by clip.
    ``on_ready(i, records)``: called per clip as its records arrive (see ``ScoringEngine.score_clips``).

    Return dtype: ``sums_only=True`` -> ``SUMS_DTYPE`` (the five sums, 40 bytes per frame: all that Content / Adaptive /
    Threshold decisions read, and all that travels from the device, through the host and over the all-gather);
    ``sums_only=False`` -> ``RECORD_DTYPE`` (with the 256-bin luma histogram).  The default ``None`` means "sums only unless
    ``flags`` asks for the luma histogram" -- whatever kind of engine is behind it, so every rank of a process group returns
    the same dtype for the same flags.

    ``downscale`` / ``interpolation``: the resize in front of the detectors -- ``"auto"`` (the reference's default pipeline,
    a factor per resolution), a number, or ``None`` (``ScoringEngine.score_clips``).

    ``hist_diff_bins``: engines that can (``ScoringEngine.score_clips``) compute HistogramDetector's ``hist_diff`` for that bin count on
    the device and return ``SUMS_DIFF_DTYPE`` for resident clips -- 48 bytes per frame instead of the 1 KiB histogram, and nothing left of
    the histogram epilogue but its decision loop; other engines ignore it and return full records (``decide`` takes either)."""
    from pyscenedetect_amd.engine import _sums_of

    if sums_only is None:
        # without a HistogramDetector nobody reads the 1 KiB luma histogram of a record
        sums_only = not (flags & _native.SCORE_LUMA_HIST)
    import inspect

    from pyscenedetect_amd.engine import downscale_size

    params = inspect.signature(engine.score_clips).parameters if hasattr(engine, "score_clips") else {}
    # does any clip get resized at all?  (an engine that packs clips but knows no downscale still packs a corpus of small clips)
    resized = downscale is not None and any(downscale_size(c.shape[1], c.shape[2], downscale)[0] > 1.0 for c in clips)
    if hasattr(engine, "score_clips") and (not resized or "downscale" in params):
        kw = {"sums_only": sums_only} if "sums_only" in params else {}
        if on_ready is not None and "on_ready" in params:
            kw["on_ready"] = on_ready
        if resized:
            kw["downscale"], kw["interpolation"] = downscale, interpolation
        if hist_diff_bins is not None and "hist_diff_bins" in params and hasattr(engine, "hist_diff_device"):
            kw["hist_diff_bins"] = hist_diff_bins
        out = engine.score_clips(clips, flags=flags, edge_kernel=edge_kernel, **kw)
        if sums_only and "sums_only" not in params:
            out = [_sums_of(r) for r in out]
        if on_ready is not None and "on_ready" not in params:
            for i, r in enumerate(out):
                on_ready(i, r)
        return out
    out = []
    for i, c in enumerate(clips):
        recs = score_clip(engine, c, flags, edge_kernel, downscale, interpolation)
        out.append(_sums_of(recs) if sums_only else recs)
        if on_ready is not None:
            on_ready(i, out[-1])
    return out


def detect_corpus(engine, clips, fps, detectors: dict, group=None, edge_kernel: int = 0, auto_downscale: bool = True,
                  downscale=1, interpolation: int = 1) -> list[dict]:
    """Detect cuts in every clip of ``clips`` (each ``uint8[n,H,W,3]``, sizes may differ).

    ``auto_downscale`` / ``downscale`` / ``interpolation`` are ``SceneManager``'s attributes of the same names
    (``scene_manager.py:282-335``) with the same defaults: by default every clip is scored as ``detect(video, detector)``
    scores it -- resized by ``compute_downscale_factor(max(frame_size))`` to about 256 pixels -- so the cut lists are those
    of the reference's default pipeline (and of its benchmark, ``benchmark/__main__.py:44-61``).  ``auto_downscale=False``
    with ``downscale=1`` scores full-resolution frames.

    With an initialised ``torch.distributed`` process group the clips are sharded over the ranks
    (``distributed.score_clips_distributed``) and every rank returns the full result."""
    flags = required_flags(detectors)
    ds = "auto" if auto_downscale else (downscale if downscale is not None and downscale > 1 else None)
    use_dist = False
    try:
        import torch.distributed as dist

        use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    except ImportError:
        pass
    if use_dist:
        from pyscenedetect_amd.distributed import score_clips_distributed

        # (with a HistogramDetector the ranks exchange the sums and hist_diff -- 48 bytes per frame -- not the 1 KiB histograms)
        bins = dict(DETECTORS["hist"][1], **(detectors["hist"] or {}))["bins"] if "hist" in detectors else None
        records = score_clips_distributed(engine, clips, flags, edge_kernel, group, downscale=ds, interpolation=interpolation, hist_diff_bins=bins)
    fps_list = fps if isinstance(fps, (list, tuple)) else [fps] * len(clips)
    sizes = [scored_size(c.shape[1], c.shape[2], ds) for c in clips]       # what the detectors see (and divide by)
    # (worker threads only where a decision is mostly native code: the histogram epilogue takes 0.2 us per frame with the GIL released;
    #  the others are microseconds per clip, and handing THEM to a pool costs a GIL hand-over per clip -- the BBC flow lost 15 %)
    # (... unless the engine computes hist_diff on the device for clips that are resident there: then that epilogue is a loop over n doubles too)
    # (the sharded path: every rank receives the sums and hist_diff of every clip, however the sending rank got them)
    device_diff = use_dist or (hasattr(engine, "hist_diff_device") and len(clips) > 0 and all(hasattr(c, "data_ptr") for c in clips))
    pool = _decide_pool() if "hist" in detectors and not device_diff else None
    if not use_dist:
        # one process: a clip is decided as soon as its records are on the host, while the batches behind it are still on the GPU --
        # and (round 6) on a worker thread, several clips at a time: the native epilogues release the GIL, and behind the default
        # downscale the kernels of a pass are as short as the histogram epilogue of its clips on one core
        result: list = [None] * len(clips)

        def decide_now(i, recs):
            if pool is None:
                result[i] = decide(recs, sizes[i][0], sizes[i][1], fps_list[i], detectors)
            else:
                result[i] = pool.submit(decide, recs, sizes[i][0], sizes[i][1], fps_list[i], detectors)

        # (a HistogramDetector's hist_diff comes from the device where the engine offers it: the histograms stay in HBM)
        bins = None
        if "hist" in detectors:
            bins = dict(DETECTORS["hist"][1], **(detectors["hist"] or {}))["bins"]
        score_clips(engine, clips, flags, edge_kernel, on_ready=decide_now, downscale=ds, interpolation=interpolation, hist_diff_bins=bins)
        return result if pool is None else [f.result() for f in result]
    if pool is None:
        return [decide(r, hw[0], hw[1], f, detectors) for r, hw, f in zip(records, sizes, fps_list)]
    return list(pool.map(lambda a: decide(a[0], a[1][0], a[1][1], a[2], detectors), zip(records, sizes, fps_list)))


_POOL = None


def _decide_pool():
    """Worker threads for the per-clip decisions of ``detect_corpus`` (``PSD_DECIDE_THREADS``: 0 or 1 = decide inline; default
    min(8, cores)); one pool per process."""
    global _POOL
    import os

    want = os.environ.get("PSD_DECIDE_THREADS")
    n = int(want) if want is not None else min(8, os.cpu_count() or 1)
    if n <= 1:
        return None
    if _POOL is None or _POOL._max_workers != n:
        from concurrent.futures import ThreadPoolExecutor

        _POOL = ThreadPoolExecutor(max_workers=n, thread_name_prefix="psd-decide")
    return _POOL
