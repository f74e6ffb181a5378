"""Multi-GPU sharding of the scoring path: one process per GPU, records all-gathered.

The reference has no distributed layer.  The pixel work shards naturally:
clips are independent, and inside a clip frame t only needs frame t-1, so a contiguous frame
range plus a ONE-FRAME HALO is self-contained.  Each rank scores its shard on its own GPU; the
only exchange is one all-gather of the per-frame score records (1064 B/frame) -- RCCL over xGMI
when the process group's backend is ``nccl``, gloo on CPU for tests.  Every rank then holds all
records and runs the (deterministic) decision epilogue, so cut lists are identical for any GPU
count.
"""

import numpy as np

from pyscenedetect_amd._native import RECORD_DTYPE, SCORE_LUMA_HIST, SUMS_DIFF_DTYPE, SUMS_DTYPE


def shard_range(n_items: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous near-equal split of ``range(n_items)``; first ``n % world`` ranks get one more."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def assign_clips(costs: list[int], world_size: int) -> list[list[int]]:
    """Greedy longest-first packing of clips (cost = frames x pixels) onto ranks."""
    loads = [0] * world_size
    out: list[list[int]] = [[] for _ in range(world_size)]
    for idx in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(idx)
        loads[r] += costs[idx]
    for lst in out:
        lst.sort()
    return out


def device_records_tensor(ptr: int, n: int, device):
    """Zero-copy ``uint8[n, 1064]`` torch view of ``n`` records sitting in HBM at ``ptr`` (``psd_last_records_device``)."""
    import torch

    class _View:
        __cuda_array_interface__ = {"shape": (n, RECORD_DTYPE.itemsize), "typestr": "|u1", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_View(), device=device)


def all_gather_records(local: np.ndarray, group=None, device_records: tuple[int, int] | None = None, counts=None) -> list[np.ndarray]:
    """All-gather ragged per-rank record arrays; returns one array per rank, in rank order.

    One padded byte tensor -- and, unless the caller knows them, the counts (tiny) in front of it.  ``counts``: the number of
    records of every rank when all ranks can work it out alike (the sharded corpus flow: the plan is deterministic and every
    clip's length is known everywhere); that saves the first collective and its device -> host synchronisation.  With the
    ``nccl`` backend the payload travels GPU-to-GPU (RCCL); the records are KBs-MBs, so this is latency-bound and a single
    fused all-gather is the cheapest pattern on the point-to-point xGMI mesh.
    ``device_records = engine.last_records_device()`` hands over the records where the kernels left them, so the
    send buffer is filled by one device-to-device copy instead of a host -> device upload;
    ``local`` (the host copy the engine returned anyway) is then only used for its length.
    """
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local)
    dtype = local.dtype     # RECORD_DTYPE, SUMS_DTYPE (records without the histogram: 40 B/frame) or SUMS_DIFF_DTYPE (+ hist_diff: 48); the same on every rank
    assert dtype in (RECORD_DTYPE, SUMS_DTYPE, SUMS_DIFF_DTYPE)
    world = dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    if counts is None:
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        mine = torch.tensor([len(local)], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts, mine, group=group)
        counts = counts.cpu().tolist()
    else:
        counts = [int(c) for c in counts]
        if len(counts) != world or counts[dist.get_rank(group)] != len(local):
            raise ValueError("counts must hold one entry per rank, this rank's equal to its number of records")
    cap = max(max(counts), 1) * dtype.itemsize
    if on_gpu and device_records is not None and device_records[1] == len(local) and len(local) and dtype != SUMS_DIFF_DTYPE:
        send = torch.zeros(cap, dtype=torch.uint8, device=dev)
        # (the device records are always 1064 bytes apart; sums are their first 40)
        send[: local.nbytes] = device_records_tensor(device_records[0], len(local), dev)[:, : dtype.itemsize].reshape(-1)
    else:
        send = torch.zeros(cap, dtype=torch.uint8)
        if len(local):
            send[: local.nbytes] = torch.from_numpy(local.view(np.uint8).reshape(-1))
        send = send.to(dev)
    recv = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.cpu().numpy().reshape(world, cap)
    return [recv[r, : counts[r] * dtype.itemsize].copy().view(dtype) for r in range(world)]


def score_clip_sharded(engine, get_frames, n_frames: int, flags: int, edge_kernel: int = 0, group=None) -> np.ndarray:
    """Score one long clip across all ranks by frame range (1-frame halo) and return ALL records.

    ``get_frames(start, stop)`` returns ``uint8[stop-start,H,W,3]`` host frames of this clip.
    """
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    start, stop = shard_range(n_frames, world, rank)
    local = np.zeros(0, RECORD_DTYPE)
    if stop > start:
        halo = get_frames(start - 1, start)[0] if start > 0 else None
        local = engine.score_host(get_frames(start, stop), prev=halo, flags=flags, edge_kernel=edge_kernel)
    parts = all_gather_records(local, group)   # (host frames are scored in staging chunks: no single device record buffer)
    return np.concatenate(parts) if parts else np.zeros(0, RECORD_DTYPE)


def sums_with_hist_diff(records: np.ndarray, bins: int) -> np.ndarray:
    """``SUMS_DIFF_DTYPE`` of one clip's records: as they are if the engine computed ``hist_diff`` on the device, else from full records
    through the host epilogue (``epilogue.hist_cuts``' values: the same bits) -- what the sharded flow exchanges with a HistogramDetector."""
    if records.dtype == SUMS_DIFF_DTYPE:
        return records
    from pyscenedetect_amd import epilogue

    out = np.empty(len(records), SUMS_DIFF_DTYPE)
    for name in SUMS_DTYPE.names:
        out[name] = records[name]
    out["hist_diff"] = epilogue.hist_cuts(records, 25.0, 0.2, bins, 15)[1] if len(records) else np.zeros(0)     # (the values depend on the bins only)
    return out


def score_clips_distributed(engine, clips, flags: int, edge_kernel: int = 0, group=None, downscale=None,
                            interpolation: int = 1, hist_diff_bins: int | None = None) -> list[np.ndarray]:
    """Score independent clips sharded by clip; returns every clip's records on every rank.

    ``clips`` is a list of objects with ``len()`` and ``[a:b]`` slicing to ``uint8[n,H,W,3]``
    (each rank only touches the clips assigned to it).  ``downscale`` / ``interpolation``: the resize in front of the
    detectors, as in ``corpus.score_clips`` (``"auto"``: the reference's default pipeline, a factor per resolution).
    The shards are balanced on the SOURCE pixels either way: that is what a rank reads from HBM.

    ``hist_diff_bins`` (a HistogramDetector's bin count, with ``SCORE_LUMA_HIST`` in ``flags``): the ranks exchange ``SUMS_DIFF_DTYPE`` --
    the five sums and the frame's ``hist_diff``, 48 bytes per frame -- instead of records with their 1 KiB histograms: every rank turns its
    clips' records into that form before the exchange (on the device where the engine can, ``psd_hist_diff_device``; through the host
    epilogue otherwise: the same bits), so all ranks send the same dtype whatever engine and wherever the clips sit.
    """
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    costs = [int(c.shape[0]) * int(c.shape[1]) * int(c.shape[2]) for c in clips]
    plan = assign_clips(costs, world)
    from pyscenedetect_amd.corpus import score_clips

    # this rank's clips, packed by resolution into shared device batches where the engine can (one launch per batch)
    want_diff = hist_diff_bins is not None and bool(flags & SCORE_LUMA_HIST)
    mine = score_clips(engine, [clips[i] for i in plan[rank]], flags, edge_kernel, downscale=downscale, interpolation=interpolation,
                       hist_diff_bins=hist_diff_bins if want_diff else None)
    if want_diff:
        mine = [sums_with_hist_diff(r, hist_diff_bins) for r in mine]
    # (a rank without clips must still send the dtype the others send: what corpus.score_clips returns for these flags WHATEVER
    #  engine is behind it -- sums unless the luma histogram was asked for.  Until round 6 this line also asked whether the engine
    #  packs clips, so with a clip-by-clip engine, no HistogramDetector and more ranks than clips the idle ranks offered 1064-byte
    #  records to an all-gather of 40-byte ones: found by the world-8 gloo test of the resized corpus.)
    sums = not (flags & SCORE_LUMA_HIST)
    local = np.concatenate(mine) if mine else np.zeros(0, SUMS_DIFF_DTYPE if want_diff else SUMS_DTYPE if sums else RECORD_DTYPE)
    # every rank can count every rank's records: the plan is deterministic and a clip's length is known wherever its shape is
    counts = [sum(int(clips[i].shape[0]) for i in plan[r]) for r in range(world)]
    comm = native_comm_for(engine, group)
    if comm is not None:
        # the exchange the C-ABI advertises (psd_comm_* / psd_allgather_host): RCCL loaded by libpsd_hip.so, one ncclAllGather, no
        # torch tensors staged, no counts collective
        parts = comm.all_gather_host(local, counts)
    else:
        parts = all_gather_records(local, group, counts=counts)
    out: list = [None] * len(clips)
    for r in range(world):
        off = 0
        for i in plan[r]:
            out[i] = parts[r][off: off + len(clips[i])]
            off += len(clips[i])
    return out


_native_comms: dict = {}      # (id(engine), id(group)) -> NativeComm | None: one communicator per engine and process group


def native_comm_for(engine, group=None, lib=None):
    """The ``NativeComm`` of ``engine`` over ``group``, made on first use, or ``None`` where the flow keeps to
    ``torch.distributed``: the group's backend is not ``nccl`` (CPU tests over gloo), the engine is not the HIP engine,
    ``PSD_NATIVE_EXCHANGE=0``, or RCCL could not be loaded / initialised on SOME rank (the ranks agree on the outcome through one
    all-reduce, once per group: a communicator only some ranks hold would hang the first collective).  ``lib``: tests hand in the
    CPU stand-in of the C-ABI (``oracle/libpsd_oracle_abi.so``) to run this path over gloo."""
    import os

    import torch
    import torch.distributed as dist

    key = (id(engine), id(group))
    if key in _native_comms:
        return _native_comms[key]
    comm = None
    wanted = os.environ.get("PSD_NATIVE_EXCHANGE", "1") != "0" and getattr(engine, "_h", None) is not None and \
        (lib is not None or dist.get_backend(group) == "nccl")
    if wanted:      # (every rank evaluates `wanted` alike: same environment, same kind of engine, same backend)
        try:
            comm = NativeComm.from_process_group(engine, group, lib=lib)
            ok = 1
        except Exception:  # noqa: BLE001 -- no RCCL, an initialisation error: this rank votes for the torch path
            ok = 0
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        vote = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
        if int(vote.item()) == 0:
            if comm is not None:
                comm.close()
            comm = None
    _native_comms[key] = comm
    return comm


class NativeComm:
    """The exchange step through the C-ABI itself (``psd_comm_*`` / ``psd_allgather_scores``): RCCL loaded by
    ``libpsd_hip.so``, no torch in the data path.  ``unique_id`` (128 bytes from :meth:`make_unique_id` on rank 0) reaches
    the other ranks by whatever means the host has; :meth:`from_process_group` uses an existing ``torch.distributed``
    group (any backend) just for that hand-over."""

    def __init__(self, engine, n_ranks: int, rank: int, unique_id: bytes, lib=None):
        import ctypes

        from pyscenedetect_amd import _native

        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of psd_comm_unique_id")
        self._engine, self._lib, self.n_ranks, self.rank = engine, lib if lib is not None else _native.load(), int(n_ranks), int(rank)
        self.exchanges = 0
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(unique_id, 128)
        self._check(self._lib.psd_comm_create(engine._h, self.n_ranks, self.rank, buf, ctypes.byref(h)))
        self._h = h

    def _check(self, rc: int) -> None:
        from pyscenedetect_amd import _native

        if rc != 0 and self._lib is not _native._lib:        # (a stand-in library keeps its own error text)
            msg = (self._lib.psd_last_error() or b"").decode("utf-8", "replace")
            raise (ValueError if rc == _native.PSD_ERR_INVALID else NotImplementedError if rc == _native.PSD_ERR_UNSUPPORTED else RuntimeError)(msg)
        _native.check(rc)

    @staticmethod
    def make_unique_id(lib=None) -> bytes:
        import ctypes

        from pyscenedetect_amd import _native

        buf = ctypes.create_string_buffer(128)
        if lib is None:
            _native.check(_native.load().psd_comm_unique_id(buf))
        elif lib.psd_comm_unique_id(buf) != 0:
            raise RuntimeError((lib.psd_last_error() or b"").decode("utf-8", "replace"))
        return buf.raw

    @classmethod
    def from_process_group(cls, engine, group=None, lib=None) -> "NativeComm":
        import torch.distributed as dist

        box = [cls.make_unique_id(lib) if dist.get_rank(group) == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(engine, dist.get_world_size(group), dist.get_rank(group), box[0], lib=lib)

    def all_gather_host(self, local: np.ndarray, counts) -> list[np.ndarray]:
        """Every rank's records, in rank order, from records this rank holds on the HOST (``psd_allgather_host``): what a
        corpus pass leaves -- several submissions' worth, ``SUMS_DTYPE`` or ``RECORD_DTYPE``, the same on every rank.
        ``counts[r]`` = number of records of rank r, the same list on every rank (the sharded flow derives it from its plan).
        Whatever is wrong with ``local`` on THIS rank, the rank still enters the collective (the library zero-fills) and
        raises afterwards."""
        import ctypes

        local = np.ascontiguousarray(local)
        dtype = local.dtype
        assert dtype in (RECORD_DTYPE, SUMS_DTYPE, SUMS_DIFF_DTYPE)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        if counts.shape != (self.n_ranks,):
            raise ValueError("counts must hold one entry per rank")     # (the same array everywhere: every rank raises)
        out = np.zeros(int(counts.sum()), dtype)
        rc = self._lib.psd_allgather_host(self._h, ctypes.c_void_p(local.ctypes.data if len(local) else None), len(local),
                                          ctypes.c_size_t(dtype.itemsize), ctypes.c_void_p(counts.ctypes.data),
                                          ctypes.c_void_p(out.ctypes.data if len(out) else None))
        self._check(rc)
        self.exchanges += 1
        parts, off = [], 0
        for c in counts:
            parts.append(out[off:off + int(c)])
            off += int(c)
        return parts

    def all_gather_records(self, counts, local: np.ndarray | None = None, device_records: tuple[int, int] | None = None) -> list[np.ndarray]:
        """Every rank's records, in rank order.  ``counts[r]`` = number of records of rank r (the same list on every
        rank); this rank's come from ``device_records`` (default: the engine's last collected submission, still in HBM).

        The device-resident source covers ONE submission: a rank whose share was scored in several submissions
        (``score_clips`` over several resolutions, chunked host input) passes ``device_records`` of a buffer it
        assembled itself, or uses :func:`all_gather_records` (host records).  A rank with nothing to contribute
        (``counts[rank] == 0``: more ranks than clips) does not consult the engine at all.  Whatever is wrong on THIS
        rank -- nothing collected yet, a stale submission, ``local`` of another length -- the rank still enters the
        collective (zero-filled) and raises afterwards, so the other ranks are never left waiting in ncclAllGather."""
        from pyscenedetect_amd import _native

        counts = np.ascontiguousarray(counts, dtype=np.int32)
        if counts.shape != (self.n_ranks,):
            raise ValueError("counts must hold one entry per rank")     # (the same array everywhere: every rank raises)
        mine = int(counts[self.rank])
        ptr, n, problem = None, 0, None
        if device_records is not None:
            ptr, n = device_records
        elif mine > 0:
            try:
                ptr, n = self._engine.last_records_device()
            except Exception as ex:  # noqa: BLE001 -- reported after the collective
                ptr, n, problem = None, -1, f"no device-resident records on rank {self.rank}: {ex}"
        if problem is None and local is not None and len(local) != n:
            ptr, n, problem = None, -1, f"local records ({len(local)}) and device records ({n}) disagree on rank {self.rank}"
        out = np.zeros(int(counts.sum()), RECORD_DTYPE)
        rc = self._lib.psd_allgather_scores(self._h, ptr if n > 0 else None, int(n), counts.ctypes.data,
                                            out.ctypes.data if len(out) else None)
        if problem is not None:
            raise ValueError(problem)
        _native.check(rc)
        parts, off = [], 0
        for c in counts:
            parts.append(out[off:off + int(c)])
            off += int(c)
        return parts

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.psd_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
