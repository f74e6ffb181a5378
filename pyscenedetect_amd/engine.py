"""Python face of the scoring engine: device-resident or host frame batches -> integer records.

Thin wrapper over the C-ABI (``include/psd_engine.h``); torch is optional and only used as a
device-memory owner (``data_ptr()``), never for arithmetic.
"""

import ctypes
import os
import threading

import numpy as np

from pyscenedetect_amd import _native
from pyscenedetect_amd._native import (  # noqa: F401  (re-exported)
    RECORD_DTYPE,
    SUMS_DIFF_DTYPE,
    SUMS_DTYPE,
    SCORE_ALL,
    SCORE_BYTE_SUM,
    SCORE_EDGES,
    SCORE_HSV_SAD,
    SCORE_LUMA_HIST,
)


def device_count() -> int:
    n = ctypes.c_int(0)
    _native.load().psd_device_count(ctypes.byref(n))
    return n.value


class DeviceBuffer:
    """A raw HBM allocation owned by an engine (for hosts without torch)."""

    def __init__(self, engine: "ScoringEngine", nbytes: int):
        self._engine = engine
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        _native.check(engine._lib.psd_device_alloc(engine._h, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, host: np.ndarray, offset: int = 0) -> None:
        host = np.ascontiguousarray(host)
        if offset + host.nbytes > self.nbytes:
            raise ValueError("upload out of range")
        _native.check(self._engine._lib.psd_memcpy_h2d(self._engine._h, self.ptr + offset, host.ctypes.data, host.nbytes))

    def upload_unordered(self, host: np.ndarray, offset: int = 0) -> None:
        """Blocking copy that does not wait for the engine's stream (``psd_upload``): for a decode thread filling a batch
        buffer that no queued kernel reads."""
        host = np.ascontiguousarray(host)
        if offset + host.nbytes > self.nbytes:
            raise ValueError("upload out of range")
        _native.check(self._engine._lib.psd_upload(self._engine._h, self.ptr + offset, host.ctypes.data, host.nbytes))

    def upload_rows(self, frame: np.ndarray, offset: int, rows: np.ndarray) -> None:
        """``upload_unordered`` for a frame uint8[H,W,3] of which only ``rows`` (ascending int32, from
        ``ScoringEngine.downscale_source_rows``) are needed: they go to the same rows of the packed frame at ``offset``; the
        rows in between keep what the buffer held (``psd_upload_rows``)."""
        if frame.dtype != np.uint8 or frame.ndim != 3 or frame.strides[2] != 1 or frame.strides[1] != frame.shape[2]:
            frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, c = frame.shape
        row_bytes = w * c
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        if offset + h * row_bytes > self.nbytes or (len(rows) and int(rows[-1]) >= h):
            raise ValueError("upload out of range")
        _native.check(self._engine._lib.psd_upload_rows(self._engine._h, self.ptr + offset, frame.ctypes.data, row_bytes,
                                                        frame.strides[0], rows.ctypes.data, len(rows)))

    def upload_rows_batch(self, frames, offset: int, rows: np.ndarray, frame_stride: int) -> None:
        """``upload_rows`` for a LIST of separately allocated frames uint8[H,W,3] at once (``psd_upload_rows_batch``): frame i's
        ``rows`` go to the packed device frame at ``offset + i * frame_stride``.  Worker threads of the engine gather the rows
        into page-locked memory and one asynchronous copy per call moves them; the host frames are free when the call
        returns, the device side is ordered by ``ScoringEngine.upload_fence``.  Decode thread only, one call at a time."""
        n = len(frames)
        if n == 0:
            return
        frames = [np.asarray(f) for f in frames]
        h, w, c = frames[0].shape
        row_bytes = w * c
        if any(f.shape != (h, w, c) for f in frames):
            raise ValueError("all frames of a batch must have the same size")
        stride0 = frames[0].strides[0]
        if any(f.dtype != np.uint8 or f.strides[2] != 1 or f.strides[1] != c or f.strides[0] != stride0 for f in frames) or stride0 < row_bytes:
            frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]     # mixed layouts: all packed
            stride0 = row_bytes
        ptrs = (ctypes.c_void_p * n)(*[f.ctypes.data for f in frames])
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        if offset + (n - 1) * frame_stride + h * row_bytes > self.nbytes or (len(rows) and int(rows[-1]) >= h):
            raise ValueError("upload out of range")
        _native.check(self._engine._lib.psd_upload_rows_batch(self._engine._h, self.ptr + offset, int(frame_stride), ptrs, n, row_bytes,
                                                              stride0, rows.ctypes.data, len(rows)))

    def download(self, nbytes: int | None = None, offset: int = 0) -> np.ndarray:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes, np.uint8)
        _native.check(self._engine._lib.psd_memcpy_d2h(self._engine._h, out.ctypes.data, self.ptr + offset, nbytes))
        return out

    def free(self) -> None:
        if self.ptr and self._engine._h:
            self._engine._lib.psd_device_free(self._engine._h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TapRowPolicy:
    """Which rows of a frame a host has to upload when everything is computed from the DOWNSCALED frame
    (reference scene_manager.py:666-678): a downscale reads 2 * dst_h of src_h rows (INTER_LINEAR; dst_h for NEAREST), and
    only those cross PCIe, into their own places of the full-size device frame.  Not worth the strided copies when most rows
    are needed anyway (INTER_AREA, factors below ~3), or when the rows do not fall into a few strided copies
    (``psd_upload_rows_plan``: two per frame for 1080p -> 256 x 144, dozens for a factor like 4.3).  Needs
    ``downscale_source_rows`` and ``upload_rows_plan`` of the class it is mixed into."""

    ROWS_ONLY_BELOW = 0.6
    MAX_ROW_COPIES = 8

    def tap_rows(self, height: int, width: int, downscale: float, interpolation: int = 1):
        """Ascending int32 rows to upload (``DeviceBuffer.upload_rows``), or None for whole frames."""
        if downscale <= 1.0:
            return None
        cache = self.__dict__.setdefault("_tap_rows", {})
        key = (height, width, float(downscale), int(interpolation), self.ROWS_ONLY_BELOW, self.MAX_ROW_COPIES)
        if key not in cache:
            dst_w, dst_h = max(1, round(width / downscale)), max(1, round(height / downscale))   # as analyze_device
            try:
                rows = self.downscale_source_rows(height, width, dst_h, dst_w, interpolation)
            except NotImplementedError:
                rows = None      # a mode the device refuses (PSD_ERR_UNSUPPORTED): the scoring call reports it
            if rows is not None and (len(rows) >= self.ROWS_ONLY_BELOW * height or len(self.upload_rows_plan(rows)) > self.MAX_ROW_COPIES):
                rows = None
            cache[key] = rows
        return cache[key]


class ScoringEngine(TapRowPolicy):
    """One HIP device + stream.  Not thread-safe per instance; use one engine per thread/GPU."""

    def __init__(self, device: int = 0):
        self._lib = _native.load()
        h = ctypes.c_void_p()
        rc = self._lib.psd_create(int(device), ctypes.byref(h))
        if rc == _native.PSD_ERR_NO_DEVICE:
            raise RuntimeError(
                "pyscenedetect_amd needs an AMD GPU (MI355X / gfx950): " + _native.last_error() + ". There is no CPU fallback."
            )
        _native.check(rc)
        self._h = h
        self.device = int(device)
        # the composite calls below (upload + resize + score + collect) keep engine state between their steps; the lock
        # makes an engine that ends up shared between threads safe, one at a time (distinct engines run concurrently)
        self._lock = threading.RLock()
        self._pinned: list[int] = []
        self.kernel_ms_acc = 0.0

    def close(self) -> None:
        """Destroy the engine (idempotent).  Waits for a composite call another thread may be inside (``_lock``); buffers
        and pinned memory handed out by this engine must not be used afterwards."""
        lock = self.__dict__.get("_lock")
        if lock is None:                       # __init__ failed before the engine existed
            return
        with lock:
            for buf in self.__dict__.pop("_scratch_bufs", {}).values():
                buf.free()
            if getattr(self, "_h", None):
                for p in self.__dict__.pop("_pinned", []):
                    self._lib.psd_host_free(self._h, p)
                self._lib.psd_destroy(self._h)
                self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- host frames -----------------------------------------------------------------------
    def score_host(self, frames: np.ndarray, prev: np.ndarray | None = None, flags: int = SCORE_ALL & ~SCORE_EDGES,
                   edge_kernel: int = 0, downscale: float = 1.0, interpolation: int = 1) -> np.ndarray:
        """Score ``frames`` uint8[N,H,W,3] (BGR, any row/frame strides) held in host memory."""
        if downscale > 1.0:
            return self._score_host_downscaled(frames, prev, flags, edge_kernel, downscale, interpolation)
        frames = np.asarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be uint8[N,H,W,3]")
        if frames.strides[3] != 1 or frames.strides[2] != 3:
            frames = np.ascontiguousarray(frames)
        n, h, w, _ = frames.shape
        pp = None
        if prev is not None:
            prev = np.asarray(prev)
            if prev.dtype != np.uint8 or prev.shape != (h, w, 3):
                raise ValueError("prev must be uint8[H,W,3] of the same size")
            if prev.strides != (frames.strides[1], 3, 1):
                tmp = np.empty((h, frames.strides[1]), np.uint8)
                tmp[:, : w * 3] = prev.reshape(h, w * 3)
                prev = tmp
            pp = prev.ctypes.data
        out = np.zeros(n, RECORD_DTYPE)
        if n == 0:
            return out
        _native.check(
            self._lib.psd_score_batch(self._h, frames.ctypes.data, n, h, w, frames.strides[1], frames.strides[0], pp,
                                      int(flags), int(edge_kernel), out.ctypes.data)
        )
        return out

    def _score_host_downscaled(self, frames, prev, flags, edge_kernel, factor, interpolation=1):
        """Upload, ``cv2.resize(INTER_LINEAR)`` on the device to the reference's target size
        (``scene_manager.py:670-678``), then score the small frames.  ``prev`` is resized too."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be uint8[N,H,W,3]")
        n, h, w, _ = frames.shape
        dw, dh = max(1, round(w / factor)), max(1, round(h / factor))
        out = np.zeros(n, RECORD_DTYPE)
        if n == 0:
            return out
        sstride = h * w * 3
        dstride = (dh * dw * 3 + 15) & ~15
        per_chunk = max(1, min(n, (256 << 20) // sstride))
        src = self._scratch("rs_src", (per_chunk + 1) * sstride)
        dst = self._scratch("rs_dst", (per_chunk + 1) * dstride)
        last = None if prev is None else np.ascontiguousarray(prev, dtype=np.uint8)
        rows = self.tap_rows(h, w, factor, interpolation)    # only the rows with taps travel (None: whole frames)
        done = 0
        while done < n:
            cnt = min(per_chunk, n - done)
            p = 0
            if last is not None:
                if rows is not None:
                    src.upload_rows(last, 0, rows)
                else:
                    src.upload(last.reshape(-1), 0)
                p = 1
            if rows is not None:
                # the chunk as one tall frame of cnt * h rows: where the row pattern continues across frames (1080 rows, steps
                # of 15) the whole chunk is two strided copies
                tall = (rows[None, :] + (np.arange(cnt, dtype=np.int32) * h)[:, None]).reshape(-1)
                src.upload_rows(frames[done:done + cnt].reshape(cnt * h, w, 3), p * sstride, tall)
            else:
                src.upload(frames[done:done + cnt].reshape(-1), p * sstride)
            self.resize_device(src.ptr, cnt + p, h, w, dst.ptr, dh, dw, dst_frame_stride=dstride, interpolation=interpolation)
            out[done:done + cnt] = self.score_device(dst.ptr + p * dstride, cnt, dh, dw, dw * 3, dstride,
                                                     d_prev=dst.ptr if p else None, flags=flags, edge_kernel=edge_kernel)
            last = frames[done + cnt - 1]
            done += cnt
        return out

    def score_frames(self, frames, prev=None, flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0,
                     downscale: float = 1.0, interpolation: int = 1) -> np.ndarray:
        """Score a list of separately allocated host frames (what a decoder hands out) without first
        stacking them on the host: every frame is uploaded straight into one device batch."""
        return self.analyze_frames(frames, prev, flags, edge_kernel, downscale, interpolation=interpolation)[0]

    def analyze_frames(self, frames, prev=None, flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0,
                       downscale: float = 1.0, hash_size: int = 0, interpolation: int = 1):
        """One upload of a list of host frames, then (records, thumbs): the score records for ``flags``
        (None if 0) and the grey ``hash_size`` x ``hash_size`` INTER_AREA thumbnails HashDetector needs (None if
        0), both computed from the (optionally downscaled) frames resident in HBM."""
        n = len(frames)
        if n == 0:
            return (np.zeros(0, RECORD_DTYPE) if flags else None,
                    np.zeros((0, hash_size, hash_size), np.uint8) if hash_size else None)
        first = np.asarray(frames[0])
        if first.dtype != np.uint8 or first.ndim != 3 or first.shape[2] != 3:
            raise ValueError("frames must be uint8[H,W,3]")
        h, w, _ = first.shape
        sstride = (h * w * 3 + 15) & ~15
        p = 0 if prev is None else 1
        src = self._scratch("fr_src", (n + 1) * sstride)
        items = ([prev] if p else []) + list(frames)
        rows = self.tap_rows(h, w, downscale, interpolation)     # behind a downscale only the rows with taps travel
        if rows is not None:
            # gathered by the engine's worker threads, one asynchronous copy per 32 frames, ordered in front of the resize
            self.synchronize()                                   # (nothing queued may still read the buffer)
            items = [np.asarray(f) for f in items]
            if any(f.shape != (h, w, 3) for f in items):
                raise ValueError("all frames of a batch must have the same size")
            for a in range(0, len(items), 32):
                src.upload_rows_batch(items[a:a + 32], a * sstride, rows, sstride)
            self.upload_fence()
        else:
            for i, f in enumerate(items):
                f = np.ascontiguousarray(f, dtype=np.uint8)
                if f.shape != (h, w, 3):
                    raise ValueError("all frames of a batch must have the same size")
                src.upload(f.reshape(-1), i * sstride)
        buf, fh, fw, stride = src, h, w, sstride
        if downscale > 1.0:
            fw, fh = max(1, round(w / downscale)), max(1, round(h / downscale))
            stride = (fh * fw * 3 + 15) & ~15
            buf = self._scratch("fr_dst", (n + 1) * stride)
            self.resize_device(src.ptr, n + p, h, w, buf.ptr, fh, fw, src_frame_stride=sstride, dst_frame_stride=stride,
                               interpolation=interpolation)
        records = thumbs = None
        if flags:
            records = self.score_device(buf.ptr + p * stride, n, fh, fw, fw * 3, stride, d_prev=buf.ptr if p else None,
                                        flags=flags, edge_kernel=edge_kernel)
        if hash_size:
            thumbs = self.hash_thumbs_device(buf.ptr + p * stride, n, fh, fw, hash_size, fw * 3, stride)
        return records, thumbs

    # -- feeding frames from the host -------------------------------------------------------------
    def pinned_array(self, shape, dtype=np.uint8) -> np.ndarray:
        """A numpy array in page-locked host memory (``psd_host_alloc``): frames decoded into it cross PCIe by DMA at
        full rate, and ``upload_async`` from it does not block.  Freed when the engine is closed."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        _native.check(self._lib.psd_host_alloc(self._h, nbytes, ctypes.byref(p)))
        self._pinned.append(p.value)
        buf = (ctypes.c_uint8 * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def upload_async(self, d_dst: int, host: np.ndarray) -> None:
        """Enqueue a host -> device copy on the engine's copy stream (asynchronous for ``pinned_array`` memory)."""
        host = np.ascontiguousarray(host)
        _native.check(self._lib.psd_upload_async(self._h, d_dst, host.ctypes.data, host.nbytes))

    def upload_fence(self, wait_on_host: bool = False) -> None:
        """Order the engine's stream (or the calling thread) after every ``upload_async`` issued so far."""
        _native.check(self._lib.psd_upload_fence(self._h, 1 if wait_on_host else 0))

    def copy_d2d(self, d_dst: int, d_src: int, nbytes: int) -> None:
        _native.check(self._lib.psd_memcpy_d2d(self._h, d_dst, d_src, int(nbytes)))

    def synchronize(self) -> None:
        """Wait until the engine's stream is idle."""
        _native.check(self._lib.psd_synchronize(self._h))

    def cpus_near_gpu(self) -> list[int]:
        """CPUs of the NUMA node this engine's GPU hangs off, within the calling thread's affinity mask (``psd_cpus_near_device``);
        empty when there is nothing to steer (one node, unknown, already inside it, ``PSD_FEED_NUMA=0``).  A host that decodes on
        its own threads runs them here -- ``os.sched_setaffinity(0, engine.cpus_near_gpu())`` in the decode thread -- so that the
        frames it allocates sit next to the GPU; the engine never moves a thread it did not create."""
        buf = (ctypes.c_int * 4096)()
        n = ctypes.c_int(0)
        _native.check(self._lib.psd_cpus_near_device(self._h, buf, 4096, ctypes.byref(n)))
        return [int(buf[i]) for i in range(min(n.value, 4096))]

    def downscale_source_rows(self, height: int, width: int, dst_h: int, dst_w: int, interpolation: int = 1) -> np.ndarray:
        """The source rows the device downscale of this shape and mode reads (ascending int32): all a host feeder has to
        upload when every consumer sees the downscaled frame (``psd_resize_source_rows``)."""
        rows = np.empty(height, np.int32)
        n = ctypes.c_int(0)
        _native.check(self._lib.psd_resize_source_rows(int(height), int(width), int(dst_h), int(dst_w), int(interpolation),
                                                       rows.ctypes.data, ctypes.addressof(n)))
        return rows[: n.value].copy()

    def upload_rows_plan(self, rows: np.ndarray, packed: bool = True) -> np.ndarray:
        """The strided copies ``DeviceBuffer.upload_rows`` issues for ``rows``: int32[k,4] of (first row, rows per group,
        distance between groups, groups) -- ``psd_upload_rows_plan``."""
        rows = np.ascontiguousarray(rows, np.int32)
        n = ctypes.c_int(0)
        _native.check(self._lib.psd_upload_rows_plan(rows.ctypes.data, len(rows), int(packed), None, 0, ctypes.addressof(n)))
        plan = np.zeros((n.value, 4), np.int32)
        _native.check(self._lib.psd_upload_rows_plan(rows.ctypes.data, len(rows), int(packed), plan.ctypes.data, n.value,
                                                     ctypes.addressof(n)))
        return plan

    def analyze_device(self, d_frames: int, n: int, height: int, width: int, frame_stride: int, d_prev: int | None = None,
                       flags: int = 0, edge_kernels=(0,), downscale: float = 1.0, hash_sizes=(), interpolation: int = 1,
                       want_frames: bool = False) -> dict:
        """Everything a batch of resident frames has to yield for a set of detectors, from ONE copy of the batch in HBM:
        ``records`` (score records for ``flags``; ``edge_xor`` for the first entry of ``edge_kernels``), ``edge_xor``
        ({kernel: array} when detectors use different dilation sizes), ``thumbs`` ({size: uint8[n,size,size]}) and, if
        ``want_frames``, ``frames``: the frames exactly as the reference's detectors and callbacks would see them
        (downscaled like ``scene_manager.py:666-678``).  ``size`` is that (height, width)."""
        with self._lock:
            kernels = list(dict.fromkeys(edge_kernels)) or [0]
            fh, fw = height, width
            ptr, stride, prev = d_frames, frame_stride, d_prev
            out = {"records": None, "edge_xor": {}, "thumbs": {}, "frames": None}
            small = None
            if downscale > 1.0:
                fw, fh = max(1, round(width / downscale)), max(1, round(height / downscale))
                if flags and not hash_sizes and not want_frames and len(kernels) == 1:
                    # resize + score inside the engine (one fused kernel for the HSV term behind INTER_LINEAR)
                    out["records"] = self.score_device_downscaled(d_frames, n, height, width, fh, fw, frame_stride, d_prev, flags,
                                                                  kernels[0], interpolation)
                    out["size"] = (fh, fw)
                    return out
                stride = (fh * fw * 3 + 15) & ~15
                small = self._scratch("an_small", (n + 1) * stride)
                if d_prev:
                    self.resize_device(d_prev, 1, height, width, small.ptr, fh, fw, src_frame_stride=frame_stride,
                                       dst_frame_stride=stride, interpolation=interpolation)
                self.resize_device(d_frames, n, height, width, small.ptr + stride, fh, fw, src_frame_stride=frame_stride,
                                   dst_frame_stride=stride, interpolation=interpolation)
                ptr, prev = small.ptr + stride, (small.ptr if d_prev else None)
            out["size"] = (fh, fw)
            if flags:
                out["records"] = self.score_device(ptr, n, fh, fw, fw * 3, stride, d_prev=prev, flags=flags, edge_kernel=kernels[0])
                if flags & SCORE_EDGES:
                    out["edge_xor"][kernels[0]] = out["records"]["edge_xor"]
                    for k in kernels[1:]:
                        out["edge_xor"][k] = self.score_device(ptr, n, fh, fw, fw * 3, stride, d_prev=prev, flags=SCORE_EDGES,
                                                               edge_kernel=k)["edge_xor"]
            for size in hash_sizes:
                out["thumbs"][size] = self.hash_thumbs_device(ptr, n, fh, fw, size, fw * 3, stride)
            if want_frames and small is not None:
                raw = small.download(n * stride, offset=stride).reshape(n, stride)
                out["frames"] = raw[:, : fh * fw * 3].reshape(n, fh, fw, 3)
            return out

    def downscale_host(self, frames: np.ndarray, downscale: float, interpolation: int = 1) -> np.ndarray:
        """Host frames uint8[N,H,W,3] as the reference's detectors and callbacks would be handed them behind a downscale
        (``cv2.resize`` of ``scene_manager.py:666-678``), made on the device: one upload, ``analyze_device(want_frames=True)``."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        if frames.ndim != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be uint8[N,H,W,3]")
        n, h, w, _ = frames.shape
        if n == 0 or downscale <= 1.0:
            return frames.copy()
        stride = (h * w * 3 + 15) & ~15
        with self._lock:
            src = self._scratch("ds_src", n * stride)
            for i in range(n):
                src.upload(frames[i].reshape(-1), i * stride)
            out = self.analyze_device(src.ptr, n, h, w, stride, flags=0, downscale=downscale, interpolation=interpolation,
                                      want_frames=True)
            return np.array(out["frames"])

    # -- HashDetector thumbnails ---------------------------------------------------------------
    def hash_thumbs_device(self, d_frames: int, n: int, height: int, width: int, size: int,
                           row_stride: int | None = None, frame_stride: int | None = None) -> np.ndarray:
        """``cv2.resize(cv2.cvtColor(f, BGR2GRAY), (size, size), INTER_AREA)`` for n frames resident in HBM."""
        row_stride = width * 3 if row_stride is None else row_stride
        frame_stride = height * row_stride if frame_stride is None else frame_stride
        out = np.zeros((n, size, size), np.uint8)
        _native.check(self._lib.psd_hash_thumbs_device(self._h, d_frames, int(n), int(height), int(width), row_stride,
                                                       frame_stride, int(size), out.ctypes.data if n else None))
        return out

    def hash_bits_device(self, d_frames: int, n: int, height: int, width: int, size: int, hash_size: int,
                         row_stride: int | None = None, frame_stride: int | None = None, want_thumbs: bool = False):
        """``HashDetector.hash_frame`` as a whole for n frames resident in HBM (``psd_hash_bits_device``): thumbnails, DCT, median
        and bits on the device.  Returns ``uint8[n, hash_size**2]`` of 0 / 1 -- what ``epilogue.hash_bits(hash_thumbs_device(...))``
        returns -- or ``(bits, thumbs)``.  ``NotImplementedError`` for transforms that do not fit a workgroup's LDS (size > 64 or so):
        use the two-step form then."""
        row_stride = width * 3 if row_stride is None else row_stride
        frame_stride = height * row_stride if frame_stride is None else frame_stride
        bits = np.zeros((n, hash_size * hash_size), np.uint8)
        thumbs = np.zeros((n, size, size), np.uint8) if want_thumbs else None
        _native.check(self._lib.psd_hash_bits_device(self._h, d_frames, int(n), int(height), int(width), row_stride, frame_stride, int(size),
                                                     int(hash_size), bits.ctypes.data if n else None,
                                                     thumbs.ctypes.data if (want_thumbs and n) else None))
        return (bits, thumbs) if want_thumbs else bits

    def hash_thumbs_host(self, frames: np.ndarray, size: int, downscale: float = 1.0, interpolation: int = 1) -> np.ndarray:
        """Same for frames uint8[N,H,W,3] in host memory (staged in bounded chunks by the engine)."""
        frames = np.asarray(frames)
        if frames.dtype != np.uint8 or frames.ndim != 4 or frames.shape[3] != 3:
            raise ValueError("frames must be uint8[N,H,W,3]")
        if downscale > 1.0:
            return self.analyze_frames(list(frames), None, 0, 0, downscale, size, interpolation)[1]
        if frames.strides[3] != 1 or frames.strides[2] != 3:
            frames = np.ascontiguousarray(frames)
        n, h, w, _ = frames.shape
        if size <= 0:
            raise ValueError("thumbnail size must be positive")
        out = np.zeros((n, size, size), np.uint8)
        if n == 0:
            return out
        _native.check(self._lib.psd_hash_thumbs(self._h, frames.ctypes.data, n, h, w, frames.strides[1], frames.strides[0],
                                                int(size), out.ctypes.data if n else None))
        return out

    def _scratch(self, name: str, nbytes: int) -> "DeviceBuffer":
        cache = self.__dict__.setdefault("_scratch_bufs", {})
        buf = cache.get(name)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                buf.free()
            buf = cache[name] = DeviceBuffer(self, nbytes)
        return buf

    def resize_device(self, d_src: int, n: int, src_h: int, src_w: int, d_dst: int, dst_h: int, dst_w: int,
                      src_frame_stride: int | None = None, dst_frame_stride: int | None = None, stream: int | None = None,
                      interpolation: int = 1):
        """``cv2.resize(frame, (dst_w, dst_h), interpolation=...)`` for n packed BGR frames in HBM; ``interpolation``
        takes cv2's values (0 NEAREST, 1 LINEAR, 3 AREA)."""
        _native.check(self._lib.psd_resize_device(
            self._h, d_src, int(n), int(src_h), int(src_w), src_h * src_w * 3 if src_frame_stride is None else src_frame_stride,
            d_dst, int(dst_h), int(dst_w), dst_h * dst_w * 3 if dst_frame_stride is None else dst_frame_stride,
            int(interpolation), stream))

    # -- device frames ---------------------------------------------------------------------
    def score_device(self, d_frames: int, n: int, height: int, width: int, row_stride: int | None = None,
                     frame_stride: int | None = None, d_prev: int | None = None,
                     flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0, stream: int | None = None,
                     sums_only: bool = False) -> np.ndarray:
        """Score ``n`` frames resident in HBM at device address ``d_frames`` (synchronous)."""
        self.submit_device(d_frames, n, height, width, row_stride, frame_stride, d_prev, flags, edge_kernel, stream)
        return self.collect(n, sums_only)

    def submit_device(self, d_frames: int, n: int, height: int, width: int, row_stride: int | None = None,
                      frame_stride: int | None = None, d_prev: int | None = None,
                      flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0, stream: int | None = None) -> None:
        row_stride = width * 3 if row_stride is None else row_stride
        frame_stride = height * row_stride if frame_stride is None else frame_stride
        _native.check(
            self._lib.psd_score_submit_device(self._h, d_frames, int(n), int(height), int(width), row_stride, frame_stride,
                                              d_prev, int(flags), int(edge_kernel), stream)
        )

    def score_device_segments(self, d_frames: int, n: int, height: int, width: int, seg_first, row_stride: int | None = None,
                              frame_stride: int | None = None, flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0,
                              stream: int | None = None, sums_only: bool = False) -> np.ndarray:
        """Score ``n`` resident frames that are SEVERAL clips packed back to back: ``seg_first`` holds the batch index of
        every clip's first frame (such a frame has no predecessor).  One launch per term for all clips."""
        self.submit_device_segments(d_frames, n, height, width, seg_first, row_stride, frame_stride, flags, edge_kernel, stream)
        return self.collect(n, sums_only)

    def submit_device_segments(self, d_frames: int, n: int, height: int, width: int, seg_first, row_stride: int | None = None,
                               frame_stride: int | None = None, flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0,
                               stream: int | None = None) -> None:
        """Asynchronous half of :meth:`score_device_segments`; pair with :meth:`collect` (at most ``MAX_INFLIGHT`` pending)."""
        row_stride = width * 3 if row_stride is None else row_stride
        frame_stride = height * row_stride if frame_stride is None else frame_stride
        seg = np.ascontiguousarray(seg_first, dtype=np.int32)
        _native.check(self._lib.psd_score_segments_submit_device(self._h, d_frames, int(n), int(height), int(width), row_stride,
                                                                 frame_stride, seg.ctypes.data if len(seg) else None, len(seg),
                                                                 int(flags), int(edge_kernel), stream))

    def submit_device_segments_downscaled(self, d_frames: int, n: int, src_h: int, src_w: int, dst_h: int, dst_w: int, seg_first,
                                          frame_stride: int | None = None, flags: int = SCORE_HSV_SAD, edge_kernel: int = 0,
                                          interpolation: int = 1, stream: int | None = None) -> None:
        """SEVERAL clips packed back to back (``seg_first``: the batch index of every clip's first frame), each frame resized
        to ``(dst_w, dst_h)`` as the reference's ``SceneManager`` does by default and then scored
        (``psd_score_segments_downscaled_submit_device``); pair with :meth:`collect`."""
        frame_stride = src_h * src_w * 3 if frame_stride is None else frame_stride
        seg = np.ascontiguousarray(seg_first, dtype=np.int32)
        _native.check(self._lib.psd_score_segments_downscaled_submit_device(
            self._h, d_frames, int(n), int(src_h), int(src_w), frame_stride, seg.ctypes.data if len(seg) else None, len(seg),
            int(dst_h), int(dst_w), int(interpolation), int(flags), int(edge_kernel), stream))

    def score_device_segments_downscaled(self, d_frames: int, n: int, src_h: int, src_w: int, dst_h: int, dst_w: int, seg_first,
                                         frame_stride: int | None = None, flags: int = SCORE_HSV_SAD, edge_kernel: int = 0,
                                         interpolation: int = 1, stream: int | None = None, sums_only: bool = False) -> np.ndarray:
        self.submit_device_segments_downscaled(d_frames, n, src_h, src_w, dst_h, dst_w, seg_first, frame_stride, flags, edge_kernel,
                                               interpolation, stream)
        return self.collect(n, sums_only)

    def score_clips(self, clips, flags: int = SCORE_ALL & ~SCORE_EDGES, edge_kernel: int = 0,
                    max_batch_bytes: int = 4 << 30, sums_only: bool = False, on_ready=None, downscale=None,
                    interpolation: int = 1, hist_diff_bins: int | None = None) -> list[np.ndarray]:
        """Records of many clips (host ``uint8[n,H,W,3]`` arrays, or device tensors with ``data_ptr()``): clips of one
        resolution share device batches of up to ``max_batch_bytes``, scored with ONE launch per term per batch
        (``psd_score_segments_device``) instead of one per clip -- thousands of short clips are launch-bound otherwise.
        Batches of resident clips are all submitted before the first is collected (up to ``MAX_INFLIGHT`` at a time), so
        the host-side copy of one batch's records overlaps the kernels of the next.
        The result equals ``[score_host(c) for c in clips]`` (``sums_only``: without the histogram, ``SUMS_DTYPE``).
        ``on_ready(i, records)`` is called for every clip as soon as its records are on the host -- while later batches
        are still being scored -- so the caller's decisions overlap the remaining kernels.

        ``downscale``: what the reference's ``SceneManager`` puts in front of its detectors (``scene_manager.py:110,123-140,
        666-678``) -- ``"auto"`` = ``auto_downscale`` (the default of ``detect()``: every resolution gets its own factor,
        ``compute_downscale_factor(max(width, height))``), a number = ``SceneManager.downscale``, ``None`` / 1 = none.  The
        records are then those of the RESIZED frames (``psd_score_segments_downscaled_device``; ``downscale_size`` gives
        their size), i.e. ``[score_host(c, downscale=factor) for c in clips]``.

        ``hist_diff_bins`` (with ``PSD_SCORE_LUMA_HIST`` in ``flags``): HistogramDetector's ``hist_diff`` for that bin count is computed on
        the device from the records while they are still in HBM (``psd_hist_diff_device``: bit for bit what ``epilogue.hist_cuts`` computes
        from the histograms) and the result is ``SUMS_DIFF_DTYPE`` -- the five sums and ``hist_diff`` (NaN at a clip's first frame), 48 bytes
        per frame instead of 1064 -- for resident clips; clips in host memory come back as full records."""
        want_diff = hist_diff_bins is not None and bool(flags & SCORE_LUMA_HIST)
        if want_diff:
            sums_only = False        # (host clips: full records; resident ones: sums + hist_diff)
        out: list = [None] * len(clips)
        ready = on_ready if on_ready is not None else (lambda i, r: None)
        dtype = SUMS_DIFF_DTYPE if want_diff else SUMS_DTYPE if sums_only else RECORD_DTYPE
        groups: dict[tuple[int, int], list[int]] = {}
        for i, c in enumerate(clips):
            if len(c.shape) != 4 or c.shape[3] != 3:
                raise ValueError("clips must be uint8[n,H,W,3]")
            if c.shape[0] == 0:
                out[i] = np.zeros(0, dtype)
                ready(i, out[i])
                continue
            groups.setdefault((int(c.shape[1]), int(c.shape[2])), []).append(i)
        # device clips that already sit back to back in HBM are scored in place: one job per contiguous run
        jobs: list[tuple[list[int], int, int]] = []
        for (h, w), idxs in groups.items():
            stride = h * w * 3
            run: list[int] = []
            for i in [i for i in idxs if hasattr(clips[i], "data_ptr")] + [None]:
                if i is not None and run and clips[run[-1]].data_ptr() + clips[run[-1]].shape[0] * stride == clips[i].data_ptr():
                    run.append(i)
                    continue
                if run:
                    jobs.append((run, h, w))
                run = [i] if i is not None else []
        # Every run is one submission, except the last, which ends in a tail piece of its own (_plan_pieces): the records of
        # everything in front of the tail reach the host -- and the caller's on_ready decides those clips -- while the tail is
        # on the GPU, so the host's share of a pass (0.5 - 0.9 ms for the 11 clips of the BBC stand-in) hides behind a kernel
        # instead of following the last one.  A piece that starts inside a clip starts one frame early: that frame is flagged
        # as a clip start, gives the piece's first real frame its predecessor, and its own record is dropped.
        pieces = []          # (run index, device pointer, frames, clip starts inside, first frame of the run it covers, drop_first)
        runs = []
        for ji, (run, h, w) in enumerate(jobs):
            stride = h * w * 3
            first = np.cumsum([0] + [clips[j].shape[0] for j in run[:-1]]).astype(np.int64)
            total = int(first[-1] + clips[run[-1]].shape[0])
            base = clips[run[0]].data_ptr()
            cuts = _plan_pieces(total, first, stride, last_run=ji == len(jobs) - 1 and on_ready is not None)
            if len(cuts) == 3 and os.environ.get("PSD_CLIPS_TAIL_MB") is None:
                # a cut buys the host's time for the clips in front of it (50 - 100 us of decisions each) and costs one more
                # launch's ramp, tail and halo frames (0.1 - 0.2 ms; more when the tail is a short walk): worth it from about
                # six clips and a tail of 512 frames on (BBC stand-in, 11 clips: +6 ... +9 %; the mixed corpus, 3 + 1 clips
                # with a 4K tail of 100 frames: -3 %, profiles/r05_e_*)
                ahead = sum(len(r["run"]) for r in runs) + int(np.sum(first + [clips[j].shape[0] for j in run] <= cuts[1]))
                if ahead < 6 or total - cuts[1] < 512:
                    cuts = [0, total]
            runs.append({"run": run, "first": first, "total": total, "recs": None, "done": 0, "next_clip": 0})
            starts = set(first.tolist())
            for a, b in zip(cuts[:-1], cuts[1:]):
                inside = a > 0 and a not in starts
                a0 = a - 1 if inside else a
                seg = [0] + [int(f - a0) for f in first if a < f < b]
                pieces.append((len(runs) - 1, base + a0 * stride, b - a0, seg, a, inside, h, w))
        small = {hw: downscale_size(hw[0], hw[1], downscale) for hw in groups}     # (factor, dst_h, dst_w) per resolution
        in_flight: list = []
        k = 0

        def retire_all():
            for pc in in_flight:      # something failed: retire what is still in flight, so that the engine stays usable
                try:
                    self.collect(pc[2], sums_only)
                except Exception:  # noqa: BLE001
                    pass
            in_flight.clear()

        try:
            while k < len(pieces) or in_flight:
                while k < len(pieces) and len(in_flight) < _native.MAX_INFLIGHT:
                    pc = pieces[k]
                    factor, dh, dw = small[(pc[6], pc[7])]
                    if factor > 1.0:
                        self.submit_device_segments_downscaled(pc[1], pc[2], pc[6], pc[7], dh, dw, pc[3], flags=flags,
                                                               edge_kernel=edge_kernel, interpolation=interpolation)
                    else:
                        self.submit_device_segments(pc[1], pc[2], pc[6], pc[7], pc[3], flags=flags, edge_kernel=edge_kernel)
                    in_flight.append(pc)
                    k += 1
                pc = in_flight.pop(0)
                ri, _, cnt, seg, a, inside, _, _ = pc
                if want_diff:
                    # the sums travel; the histograms stay in HBM and give hist_diff there (a clip's first frame has no predecessor)
                    sums = self.collect(cnt, True)
                    recs = np.empty(cnt, SUMS_DIFF_DTYPE)
                    for name in SUMS_DTYPE.names:
                        recs[name] = sums[name]
                    d_recs, n_dev = self.last_records_device()
                    assert n_dev == cnt
                    diff = self.hist_diff_device(d_recs, cnt, hist_diff_bins)
                    diff[np.asarray(seg, dtype=np.int64)] = np.nan
                    recs["hist_diff"] = diff
                else:
                    recs = self.collect(cnt, sums_only)
                r = runs[ri]
                if inside:
                    recs = recs[1:]
                if r["recs"] is None:
                    r["recs"] = recs if len(recs) == r["total"] else np.empty(r["total"], recs.dtype)
                if r["recs"] is not recs:
                    r["recs"][a:a + len(recs)] = recs
                r["done"] = a + len(recs)
                while r["next_clip"] < len(r["run"]):
                    j = r["run"][r["next_clip"]]
                    f0 = int(r["first"][r["next_clip"]])
                    if f0 + clips[j].shape[0] > r["done"]:
                        break
                    out[j] = r["recs"][f0:f0 + clips[j].shape[0]]
                    r["next_clip"] += 1
                    ready(j, out[j])
        except BaseException:
            retire_all()
            raise
        for (h, w), idxs in groups.items():
            stride = h * w * 3
            factor, dh, dw = small[(h, w)]
            per_batch = max(1, max_batch_bytes // stride)
            host = [i for i in idxs if not hasattr(clips[i], "data_ptr")]
            k = 0
            while k < len(host):
                batch, frames = [], 0
                while k < len(host) and (not batch or frames + clips[host[k]].shape[0] <= per_batch):
                    batch.append(host[k])
                    frames += clips[host[k]].shape[0]
                    k += 1
                if len(batch) == 1 and frames > per_batch:      # one clip larger than a batch: the chunked host path
                    recs = self.score_host(np.asarray(clips[batch[0]]), flags=flags, edge_kernel=edge_kernel,
                                           downscale=factor if factor > 1.0 else 1.0, interpolation=interpolation)
                    out[batch[0]] = _sums_of(recs) if sums_only else recs
                    ready(batch[0], out[batch[0]])
                    continue
                buf = self._scratch("clips", frames * stride)
                # behind a downscale only the source rows that carry taps cross PCIe (288 of 1080 for the default 1080p -> 256 x 144),
                # like SceneManager's feeder: the rows in between keep whatever the buffer held, nothing reads them
                rows = self.tap_rows(h, w, factor, interpolation) if factor > 1.0 else None
                first, off = [], 0
                for j in batch:
                    c = np.ascontiguousarray(clips[j], dtype=np.uint8)
                    if rows is not None:
                        # the clip as one tall frame of n * h rows: where the row pattern continues across frames the whole clip is
                        # a few strided copies (_score_host_downscaled)
                        tall = (rows[None, :] + (np.arange(c.shape[0], dtype=np.int32) * h)[:, None]).reshape(-1)
                        buf.upload_rows(c.reshape(c.shape[0] * h, w, 3), off * stride, tall)
                    else:
                        buf.upload(c.reshape(-1), off * stride)
                    first.append(off)
                    off += c.shape[0]
                if factor > 1.0:
                    recs = self.score_device_segments_downscaled(buf.ptr, frames, h, w, dh, dw, first, flags=flags, edge_kernel=edge_kernel,
                                                                 interpolation=interpolation, sums_only=sums_only)
                else:
                    recs = self.score_device_segments(buf.ptr, frames, h, w, first, flags=flags, edge_kernel=edge_kernel, sums_only=sums_only)
                for j, f0 in zip(batch, first):
                    out[j] = recs[f0:f0 + clips[j].shape[0]].copy()
                    ready(j, out[j])
        return out

    def submit_device_downscaled(self, d_frames: int, n: int, src_h: int, src_w: int, dst_h: int, dst_w: int,
                                 frame_stride: int | None = None, d_prev: int | None = None,
                                 flags: int = SCORE_HSV_SAD, edge_kernel: int = 0, interpolation: int = 1,
                                 stream: int | None = None) -> None:
        """``cv2.resize(frame, (dst_w, dst_h), interpolation)`` followed by the terms of ``flags`` on the resized frames
        (the reference's default pipeline, ``scene_manager.py:666-678``); ``d_prev`` is a SOURCE-size frame.  Pair with
        :meth:`collect`."""
        frame_stride = src_h * src_w * 3 if frame_stride is None else frame_stride
        _native.check(self._lib.psd_score_downscaled_submit_device(
            self._h, d_frames, int(n), int(src_h), int(src_w), frame_stride, d_prev, int(dst_h), int(dst_w), int(interpolation),
            int(flags), int(edge_kernel), stream))

    def score_device_downscaled(self, d_frames: int, n: int, src_h: int, src_w: int, dst_h: int, dst_w: int,
                                frame_stride: int | None = None, d_prev: int | None = None, flags: int = SCORE_HSV_SAD,
                                edge_kernel: int = 0, interpolation: int = 1, stream: int | None = None) -> np.ndarray:
        self.submit_device_downscaled(d_frames, n, src_h, src_w, dst_h, dst_w, frame_stride, d_prev, flags, edge_kernel,
                                      interpolation, stream)
        return self.collect(n)

    def collect(self, n: int, sums_only: bool = False) -> np.ndarray:
        """Records of the oldest pending submission.  ``sums_only``: ``SUMS_DTYPE`` (the five sums, 40 bytes per frame)
        instead of ``RECORD_DTYPE`` (1064 with the luma histogram) -- all that ContentDetector, AdaptiveDetector and
        ThresholdDetector decide from, and all a submission without the luma terms moves to the host anyway."""
        if sums_only:
            out = np.empty(n, SUMS_DTYPE)
            _native.check(self._lib.psd_score_collect_sums(self._h, out.ctypes.data if n else None, int(n)))
        else:
            out = np.empty(n, RECORD_DTYPE)
            _native.check(self._lib.psd_score_collect(self._h, out.ctypes.data if n else None, int(n)))
        self.kernel_ms_acc += self.last_kernel_ms()[0]      # callers that time a flow of several submissions reset and read this
        return out

    def hist_diff_device(self, d_recs: int, n: int, bins: int = 128) -> np.ndarray:
        """HistogramDetector's ``hist_diff`` (``histogram_detector.py:98,156-163``: re-bin to ``bins``, ``cv2.normalize``, ``cv2.compareHist``
        CORREL) of n records that are still in HBM (``last_records_device``), frame pairs in parallel on the device with the host
        epilogue's order of operations -- the same bits as ``epilogue.hist_cuts``; element 0 is NaN (``psd_hist_diff_device``)."""
        out = np.empty(n, np.float64)
        _native.check(self._lib.psd_hist_diff_device(self._h, d_recs, int(n), int(bins), out.ctypes.data if n else None, None))
        return out

    def last_records_device(self) -> tuple[int, int]:
        """``(device pointer, n)`` of the most recently collected submission's records, still in HBM (they feed the
        RCCL all-gather of ``pyscenedetect_amd.distributed`` without a host bounce)."""
        p, n = ctypes.c_void_p(), ctypes.c_int(0)
        _native.check(self._lib.psd_last_records_device(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def last_walk_geometry(self) -> tuple[int, int]:
        """``(frames_per_chunk, n_tiles)`` of this thread's most recent time-walking launch (``psd_last_walk_geometry``): frames
        ``k * frames_per_chunk`` of the batch are where a workgroup's walk started from a re-read halo frame."""
        fpc, tiles = ctypes.c_int(0), ctypes.c_int(0)
        _native.check(self._lib.psd_last_walk_geometry(self._h, ctypes.byref(fpc), ctypes.byref(tiles)))
        return fpc.value, tiles.value

    def last_kernel_ms(self) -> tuple[float, int]:
        ms, launches = ctypes.c_float(0), ctypes.c_int(0)
        _native.check(self._lib.psd_last_kernel_ms(self._h, ctypes.byref(ms), ctypes.byref(launches)))
        return ms.value, launches.value

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def edge_map(self, d_frame: int, height: int, width: int, row_stride: int | None = None, edge_kernel: int = 0) -> np.ndarray:
        out = np.empty((height, width), np.uint8)
        _native.check(self._lib.psd_edge_map_device(self._h, d_frame, height, width, width * 3 if row_stride is None else row_stride,
                                                    int(edge_kernel), out.ctypes.data))
        return out


def downscale_size(height: int, width: int, downscale=None) -> tuple[float, int, int]:
    """``(factor, dst_h, dst_w)`` of the resize the reference's ``SceneManager`` puts in front of its detectors for frames of
    ``height x width``: ``downscale="auto"`` is ``auto_downscale=True`` (the default, what ``detect()`` and the reference's
    benchmark run: ``compute_downscale_factor(max(frame_size))``, ``scene_manager.py:123-140,525-528``), a number is
    ``SceneManager.downscale``, ``None`` / anything ``<= 1`` no resize.  The target size is ``max(1, round(size / factor))``
    per axis (``scene_manager.py:670-678``; Python's ``round``: half to even) and only applies for ``factor > 1.0``."""
    if downscale is None:
        return 1.0, int(height), int(width)
    if isinstance(downscale, str):
        if downscale != "auto":
            raise ValueError("downscale must be None, 'auto' or a number")
        from pyscenedetect_amd.scene_manager import compute_downscale_factor

        factor = compute_downscale_factor(max(int(width), int(height)))
    else:
        factor = downscale
    if not factor > 1.0:
        return 1.0, int(height), int(width)
    return float(factor), max(1, round(height / factor)), max(1, round(width / factor))


def _plan_pieces(total: int, first, frame_bytes: int, last_run: bool = True) -> list[int]:
    """Frame indices at which ``score_clips`` cuts a run of ``total`` resident frames (clip starts at ``first``) into
    submissions: ``[0, total]`` or ``[0, cut, total]``.

    Only the LAST run of a call is cut, and only once: while its tail piece (a fifth of the run, between 1.5 and 6 GiB --
    0.3 to 1.2 ms of kernel) is on the GPU, the host collects and decides everything in front of it; what is left exposed
    after the last kernel is the tail's own few clips.  (The host's share of a pass hides behind the next run's kernel for
    every run but the last.  More, equal pieces were measured too, ``profiles/r05_c_*``: every extra launch costs its ramp
    and tail -- 4 pieces of the BBC stand-in's 20 GB ran 5 % longer on the GPU than one, which ate what the overlap gave.)
    The cut snaps to a clip start within a third of the tail.  ``PSD_CLIPS_TAIL_MB`` overrides the tail's size (0: no cut)."""
    env = os.environ.get("PSD_CLIPS_TAIL_MB")
    run_bytes = total * frame_bytes
    if env is not None:
        tail_bytes = int(env) << 20
    else:
        tail_bytes = min(max(run_bytes // 5, 1536 << 20), 6144 << 20)
    tail = tail_bytes // max(1, frame_bytes)
    if not last_run or tail_bytes <= 0 or tail < 1 or total < 3 * tail:
        return [0, total]
    cut = total - tail
    starts = np.asarray(first, dtype=np.int64)
    if len(starts):
        near = int(starts[np.argmin(np.abs(starts - cut))])
        if abs(near - cut) <= max(1, tail // 3) and 0 < near < total:
            cut = near
    return [0, int(cut), total]


def _sums_of(records: np.ndarray) -> np.ndarray:
    """``SUMS_DTYPE`` copy of full records."""
    out = np.empty(len(records), SUMS_DTYPE)
    for name in SUMS_DTYPE.names:
        out[name] = records[name]
    return out


def _locked(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return fn(self, *args, **kwargs)

    return wrapper


# the synchronous calls hold the engine's lock from their first step to their last (the submit_* / collect pairs of the
# pipelined API stay the caller's to serialise)
for _name in ("score_host", "_score_host_downscaled", "analyze_frames", "hash_thumbs_device", "hash_bits_device", "hash_thumbs_host", "score_device",
              "score_device_downscaled", "score_device_segments", "score_device_segments_downscaled", "score_clips", "edge_map",
              "resize_device"):
    setattr(ScoringEngine, _name, _locked(getattr(ScoringEngine, _name)))

_default_tls = threading.local()


def default_engine(device: int | None = None) -> ScoringEngine:
    """The calling THREAD's engine for ``device`` (default: ``LOCAL_RANK`` or 0).  An engine keeps state between the
    steps of a call (record slots, staging buffers), so threads do not share one: two SceneManagers or detectors running
    on different threads each get their own stream and buffers (the reference runs one detector set per thread,
    ``benchmark/sweep.py:160-180``).  Raises without a GPU.

    Lifetime: the thread's slot and every object that cached the engine (a SceneManager, a detector) hold ordinary
    references; the engine is destroyed when the last of them lets go -- never from another thread's call to this
    function (a SceneManager built on a worker thread stays usable after that thread has ended)."""
    import os

    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        if device >= max(device_count(), 1):
            device = 0
    cache = _default_tls.__dict__.setdefault("engines", {})
    eng = cache.get(device)
    if eng is None or eng._h is None:          # (closed by its user: make a fresh one)
        eng = cache[device] = ScoringEngine(device)
    return eng


def hsv_tables() -> tuple[np.ndarray, np.ndarray]:
    s = np.zeros(256, np.int32)
    h = np.zeros(256, np.int32)
    _native.check(_native.load().psd_hsv_tables(s.ctypes.data, h.ctypes.data))
    return s, h
