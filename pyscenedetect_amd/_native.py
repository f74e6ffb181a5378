"""ctypes binding of the C-ABI in ``include/psd_engine.h`` (``libpsd_hip.so``).

There is deliberately no CPU fallback: if the HIP library is missing, or no GPU is visible when
an engine is created, this raises.  The reference has no FFI of its own; the functions bound here
replace the cv2/numpy pixel arithmetic of the four ``process_frame()`` loops (see the header).
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSD_LIB_PATH") or os.path.join(_HERE, "libpsd_hip.so")  # override: experiments only

PSD_OK = 0
PSD_ERR_INVALID = -1
PSD_ERR_NO_DEVICE = -2
PSD_ERR_HIP = -3
PSD_ERR_UNSUPPORTED = -4
PSD_ERR_NOMEM = -5

SCORE_HSV_SAD = 1
SCORE_LUMA_HIST = 2
SCORE_BYTE_SUM = 4
SCORE_EDGES = 8
SCORE_ALL = 15

MAX_INFLIGHT = 4
ABI_VERSION = 8  # PSD_ABI_VERSION of include/psd_engine.h this binding was written against

#: numpy view of ``psd_frame_scores`` (1064 bytes).
RECORD_DTYPE = np.dtype(
    [
        ("sad_h", "<u8"),
        ("sad_s", "<u8"),
        ("sad_v", "<u8"),
        ("edge_xor", "<u8"),
        ("byte_sum", "<u8"),
        ("hist", "<u4", (256,)),
    ]
)
assert RECORD_DTYPE.itemsize == 1064

#: numpy view of ``psd_frame_sums`` (40 bytes): a record without its histogram -- what ContentDetector, AdaptiveDetector and
#: ThresholdDetector decide from (``psd_score_collect_sums``).
SUMS_DTYPE = np.dtype([(name, "<u8") for name in ("sad_h", "sad_s", "sad_v", "edge_xor", "byte_sum")])
assert SUMS_DTYPE.itemsize == 40 and all(SUMS_DTYPE.fields[k][1] == RECORD_DTYPE.fields[k][1] for k in SUMS_DTYPE.names)
#: the five sums + HistogramDetector's ``hist_diff`` of the frame, computed on the device (``psd_hist_diff_device``; NaN where a frame has no
#: predecessor): what a pass with a HistogramDetector moves to the host instead of the 1 KiB histogram (``ScoringEngine.score_clips(hist_diff_bins=)``)
SUMS_DIFF_DTYPE = np.dtype([(name, "<u8") for name in SUMS_DTYPE.names] + [("hist_diff", "<f8")])
assert SUMS_DIFF_DTYPE.itemsize == 48 and all(SUMS_DIFF_DTYPE.fields[k][1] == RECORD_DTYPE.fields[k][1] for k in SUMS_DTYPE.names)


class ContentParams(ctypes.Structure):
    _fields_ = [
        ("threshold", ctypes.c_double),
        ("weights", ctypes.c_double * 4),
        ("filter_mode", ctypes.c_int),
        ("min_len_frames", ctypes.c_int64),
        ("min_len_secs", ctypes.c_double),
    ]


class AdaptiveParams(ctypes.Structure):
    _fields_ = [
        ("adaptive_threshold", ctypes.c_double),
        ("min_content_val", ctypes.c_double),
        ("window_width", ctypes.c_int),
        ("min_len_frames", ctypes.c_int64),
        ("min_len_secs", ctypes.c_double),
    ]


class HistParams(ctypes.Structure):
    _fields_ = [
        ("threshold", ctypes.c_double),
        ("bins", ctypes.c_int),
        ("min_len_frames", ctypes.c_int64),
        ("min_len_secs", ctypes.c_double),
    ]


class ThresholdParams(ctypes.Structure):
    _fields_ = [
        ("threshold", ctypes.c_int),
        ("method", ctypes.c_int),
        ("fade_bias", ctypes.c_double),
        ("add_final_scene", ctypes.c_int),
        ("min_len_frames", ctypes.c_int64),
        ("min_len_secs", ctypes.c_double),
    ]


class HashParams(ctypes.Structure):
    _fields_ = [
        ("threshold", ctypes.c_double),
        ("hash_size", ctypes.c_int),
        ("min_len_frames", ctypes.c_int64),
        ("min_len_secs", ctypes.c_double),
    ]


#: name -> (restype, argtypes); every symbol ``include/psd_engine.h`` declares.
_vp, _sz, _i, _u32, _i64, _f = (
    ctypes.c_void_p,
    ctypes.c_size_t,
    ctypes.c_int,
    ctypes.c_uint32,
    ctypes.c_int64,
    ctypes.c_float,
)
_P = ctypes.POINTER
SYMBOLS = {
    "psd_abi_version": (_i, []),
    "psd_last_error": (ctypes.c_char_p, []),
    "psd_device_count": (_i, [_P(_i)]),
    "psd_create": (_i, [_i, _P(_vp)]),
    "psd_destroy": (None, [_vp]),
    "psd_score_batch_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _u32, _i, _vp, _vp]),
    "psd_score_submit_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _u32, _i, _vp]),
    "psd_score_segments_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _i, _u32, _i, _vp, _vp]),
    "psd_score_segments_submit_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _i, _u32, _i, _vp]),
    "psd_score_collect": (_i, [_vp, _vp, _i]),
    "psd_score_collect_sums": (_i, [_vp, _vp, _i]),
    "psd_score_batch": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _vp, _u32, _i, _vp]),
    "psd_comm_unique_id": (_i, [_vp]),
    "psd_comm_create": (_i, [_vp, _i, _i, _vp, _P(_vp)]),
    "psd_comm_destroy": (None, [_vp]),
    "psd_allgather_scores": (_i, [_vp, _vp, _i, _vp, _vp]),
    "psd_allgather_host": (_i, [_vp, _vp, _i, _sz, _vp, _vp]),
    "psd_last_records_device": (_i, [_vp, _P(_vp), _P(_i)]),
    "psd_last_kernel_ms": (_i, [_vp, _P(_f), _P(_i)]),
    "psd_device_alloc": (_i, [_vp, _sz, _P(_vp)]),
    "psd_device_free": (_i, [_vp, _vp]),
    "psd_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "psd_host_alloc": (_i, [_vp, _sz, _P(_vp)]),
    "psd_host_free": (_i, [_vp, _vp]),
    "psd_upload": (_i, [_vp, _vp, _vp, _sz]),
    "psd_upload_async": (_i, [_vp, _vp, _vp, _sz]),
    "psd_resize_source_rows": (_i, [_i, _i, _i, _i, _i, _vp, _vp]),
    "psd_upload_rows": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _i]),
    "psd_upload_rows_plan": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "psd_upload_rows_batch": (_i, [_vp, _vp, _sz, _vp, _i, _sz, _sz, _vp, _i]),
    "psd_cpus_near_device": (_i, [_vp, _vp, _i, _vp]),
    "psd_last_walk_geometry": (_i, [_vp, _P(_i), _P(_i)]),
    "psd_upload_fence": (_i, [_vp, _i]),
    "psd_memcpy_d2d": (_i, [_vp, _vp, _vp, _sz]),
    "psd_synchronize": (_i, [_vp]),
    "psd_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "psd_hsv_tables": (_i, [_vp, _vp]),
    "psd_edge_map_device": (_i, [_vp, _vp, _i, _i, _sz, _i, _vp]),
    "psd_resize_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _i, _sz, _i, _vp]),
    "psd_score_downscaled_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _i, _i, _u32, _i, _vp, _vp]),
    "psd_score_downscaled_submit_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _i, _i, _u32, _i, _vp]),
    "psd_score_segments_downscaled_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _i, _i, _i, _u32, _i, _vp, _vp]),
    "psd_score_segments_downscaled_submit_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _i, _i, _i, _u32, _i, _vp]),
    "psd_resize_linear_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _vp, _i, _i, _sz, _vp]),
    "psd_hash_thumbs_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _vp]),
    "psd_hash_thumbs": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _vp]),
    "psd_hash_bits_device": (_i, [_vp, _vp, _i, _i, _i, _sz, _sz, _i, _i, _vp, _vp]),
    "psd_epilogue_content_scores": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "psd_epilogue_content_scores_sums": (_i, [_vp, _sz, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "psd_epilogue_threshold_cuts_sums": (_i, [_vp, _sz, _i, _i, _i, _i64, _i64, _i64, _P(ThresholdParams), _vp, _vp, _P(_i)]),
    "psd_epilogue_content_cuts": (_i, [_vp, _i, _i64, _i64, _i64, _P(ContentParams), _vp, _P(_i)]),
    "psd_epilogue_adaptive_cuts": (_i, [_vp, _i, _i64, _i64, _i64, _P(AdaptiveParams), _vp, _vp, _P(_i)]),
    "psd_epilogue_hist_cuts": (_i, [_vp, _i, _vp, _i64, _i64, _i64, _P(HistParams), _vp, _vp, _P(_i)]),
    "psd_epilogue_hist_cuts_from_diff": (_i, [_vp, _i, _i64, _i64, _i64, _P(HistParams), _vp, _P(_i)]),
    "psd_hist_diff_device": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "psd_epilogue_hist_normalize": (_i, [_vp, _i, _vp]),
    "psd_epilogue_hist_correl": (_i, [_vp, _vp, _i, _P(ctypes.c_double)]),
    "psd_epilogue_threshold_cuts": (_i, [_vp, _i, _i, _i, _i64, _i64, _i64, _P(ThresholdParams), _vp, _vp, _P(_i)]),
    "psd_epilogue_hash_bits": (_i, [_vp, _i, _i, _i, _vp]),
    "psd_epilogue_hash_cuts": (_i, [_vp, _i, _vp, _i64, _i64, _i64, _P(HashParams), _vp, _vp, _P(_i)]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    """libpsd_hip.so is missing or unusable.  There is no CPU fallback."""


def _preload_torch_hip_runtime() -> None:
    """Make this process use ONE HIP/HSA runtime.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so.7``/``libhsa-runtime64`` next to
    ``libtorch_hip.so``; ``libpsd_hip.so`` links against the same SONAME.  If ``/opt/rocm``'s copy
    were loaded first and torch initialised later (torch owns device memory and RCCL in
    ``bench.py``), two HSA runtimes would fight over the GPU ("No HIP GPUs are available").
    So when torch is installed, load *its* runtime first (without importing torch); the dynamic
    linker then binds our DT_NEEDED entry to it.
    """
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # torch already brought its runtime in
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pyscenedetect_amd/csrc`).  pyscenedetect_amd has no CPU fallback."
        )
    _preload_torch_hip_runtime()
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as ex:  # e.g. libamdhip64 missing
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {ex}") from ex
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as ex:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from ex
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.psd_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library has {lib.psd_abi_version()}, binding wants {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return (load().psd_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map a psd_status to the Python exception the reference would raise for the same fault."""
    if rc == PSD_OK:
        return
    msg = last_error()
    if rc == PSD_ERR_INVALID:
        raise ValueError(msg)
    if rc == PSD_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == PSD_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(f"psd error {rc}: {msg}")
