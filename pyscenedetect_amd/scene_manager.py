"""SceneManager: runs detectors over a frame source, batching the pixel work onto the GPU.

API surface follows the reference's ``scenedetect/scene_manager.py`` (``SceneManager`` :218-745,
``get_scenes_from_cuts`` :171-210, ``compute_downscale_factor`` :123-140).  The difference is
inside ``detect_scenes``: instead of calling ``detector.process_frame`` once per frame per detector
(:578-597, :426-428), frames are gathered into batches by a decode thread, each batch is scored
ONCE on the device for the union of what all registered detectors need, and every detector then
replays its decision logic over the per-frame records in frame order -- so cut lists, metrics and
callback order are those of the sequential loop.
"""

import logging
import os
import queue
from enum import Enum
import sys
import threading
import typing as ty
import warnings

import numpy as np

from pyscenedetect_amd.detector import SceneDetector
from pyscenedetect_amd.stats_manager import StatsManager
from pyscenedetect_amd.timecode import FrameTimecode, adopt, give_back

logger = logging.getLogger("pyscenedetect")

PROGRESS_BAR_DESCRIPTION = "  Detected: %d | Progress"
"""Template of the progress bar's description (reference ``scene_manager.py:119``)."""


class _NoProgress:
    """What ``show_progress=True`` gets where ``tqdm`` is not installed (reference ``platform.py:53-67``)."""

    def __init__(self, **kwargs):
        pass

    def update(self, n=1):
        pass

    def close(self):
        pass

    def set_description(self, desc=None, refresh=True):
        pass


try:
    from tqdm import tqdm as _tqdm
except ModuleNotFoundError:      # optional, like in the reference
    _tqdm = _NoProgress

DEFAULT_MIN_WIDTH: int = 256
MAX_FRAME_QUEUE_LENGTH: int = 4
MAX_FRAME_SIZE_ERRORS: int = 16
DEFAULT_BATCH_FRAMES: int = 64


class Interpolation(Enum):
    """Resize filters of the reference (``scenedetect/common.py:148-160``, values = OpenCV's ``INTER_*``), all five on the device.
    LINEAR (the reference's default), NEAREST, AREA and LANCZOS4 have ONE 8-bit result in OpenCV (integer arithmetic behind the
    coefficient tables).  CUBIC does not: its 8-bit vertical pass is float32 SIMD on most builds, fused or not, scalar fixed point on
    others, and the x86-64 PyPI wheels hand it to IPP.  The device reproduces, byte for byte, the form ``PSD_CUBIC_FORM`` names --
    ``sse`` (default: OpenCV 4.x without IPP on an SSE baseline), ``fma`` (aarch64 / FMA baselines), ``fixed`` (no vector pass);
    they differ in about one byte per 50,000."""

    NEAREST = 0
    LINEAR = 1
    CUBIC = 2
    AREA = 3
    LANCZOS4 = 4


def compute_downscale_factor(frame_width: int, effective_width: int = DEFAULT_MIN_WIDTH) -> float:
    """Downscale factor that brings ``frame_width`` to about ``effective_width`` pixels."""
    assert frame_width > 0 and effective_width > 0
    if frame_width < effective_width:
        return 1
    return frame_width / float(effective_width)


def expand_scenes_to_bounds(scenes, start, end):
    """New scene list whose first scene starts at ``start`` and whose last scene ends at ``end`` (scenes detected
    inside a sub-window of a video, reference ``scene_manager.py:143-168``).  The input is not modified."""
    expanded = list(scenes)
    if expanded:
        expanded[0] = (start, expanded[0][1])
        expanded[-1] = (expanded[-1][0], end)
    return expanded


def get_scenes_from_cuts(cut_list, start_pos, end_pos):
    """Contiguous (start, end) pairs from a sorted cut list; one scene if there are no cuts."""
    bounds = [start_pos, *cut_list, end_pos]
    return list(zip(bounds[:-1], bounds[1:]))


def _scaled_shape(shape, factor) -> tuple[int, int]:
    """(height, width) of a frame of ``shape`` as the detectors see it behind a downscale by ``factor`` (scene_manager.py:670-678)."""
    if factor > 1.0:
        return max(1, round(shape[0] / factor)), max(1, round(shape[1] / factor))
    return int(shape[0]), int(shape[1])


def _patch_first_record(engine, plan, result, seam) -> None:
    """The first record of a batch scored against ``seam[0]`` -- the previous call's last frame AS ITS detectors saw it -- where
    the two calls reach the same size by different downscales (``detect_scenes``): ``seam[1]`` is the batch's first frame as seen."""
    prev_small, cur_small = seam
    flags = plan["flags"] & 9
    if not flags or result.get("records") is None:
        return
    kernels = plan["kernels"] if plan["flags"] & 8 else [0]
    for k in kernels:
        rec = engine.score_host(cur_small[None], prev=prev_small, flags=flags, edge_kernel=k)
        if k == kernels[0]:
            for name in ("sad_h", "sad_s", "sad_v", "edge_xor"):
                result["records"][name][0] = rec[name][0]
        if plan["flags"] & 8 and k in result.get("edge_xor", {}):
            result["edge_xor"][k][0] = rec["edge_xor"][0]


def _score_flags(detector) -> int:
    """The ``PSD_SCORE_*`` terms a detector wants from the device; 0 for a detector that only knows the plug-in API (one written
    against the reference's ``scenedetect.SceneDetector`` has none of this package's extension methods)."""
    fn = getattr(detector, "score_flags", None)
    return int(fn()) if fn is not None else 0


def _edge_kernel(detector) -> int:
    fn = getattr(detector, "edge_kernel_size", None)
    return int(fn()) if fn is not None else 0


def _thumb_size(detector) -> int:
    fn = getattr(detector, "hash_thumb_size", None)
    return int(fn()) if fn is not None else 0


class SceneManager:
    def __init__(self, stats_manager: StatsManager | None = None, engine=None,
                 batch_frames: int = DEFAULT_BATCH_FRAMES):
        self._cutting_list: list[FrameTimecode] = []
        self._detector_list: list[SceneDetector] = []
        self._stats_manager = stats_manager
        if stats_manager is not None and not isinstance(stats_manager, StatsManager):
            # (the reference's StatsManager writes only rows keyed by ITS FrameTimecode class: filled by this manager it saves a header)
            logger.warning("stats_manager is a %s.%s, not a pyscenedetect_amd.StatsManager: metrics keyed by this package's timecodes "
                           "may not be saved by it.", type(stats_manager).__module__, type(stats_manager).__name__)
        self._engine = engine
        self._batch_frames = max(1, int(batch_frames))
        self._start_pos = None
        self._last_pos = None
        self._frame_size = None
        self._frame_size_errors = 0
        self._base_timecode = None
        self._downscale = 1
        self._auto_downscale = True
        self._interpolation = Interpolation.LINEAR
        self._exception_info = None
        self._stop = threading.Event()
        # (position, frame, pending): ``pending`` is None for a frame that is already what the reference's buffer would hold, or
        # (factor, interpolation) for a full-size frame of a call that needed no downscaled frames on the host (see _seen_frame)
        self._frame_buffer: list[tuple] = []
        self._frame_buffer_size = 0
        self._crop = None
        self._carry_frame = None     # the last frame of the previous detect_scenes() call (see there): this manager's own copy
        self._carry_scale = None     # ... and the (downscale factor, interpolation) it was scored behind
        self._roles: list[tuple] = []  # (detector, score flags, thumbnail size, edge kernel) per registered detector (_dispatch)

    # -- configuration (reference :265-335) ---------------------------------------------------------
    @property
    def stats_manager(self):
        return self._stats_manager

    @property
    def crop(self):
        if self._crop is None:
            return None
        x0, y0, x1, y1 = self._crop
        return (x0, y0, x1 - 1, y1 - 1)

    @crop.setter
    def crop(self, value):
        if value is None:
            self._crop = None
            return
        if not (len(value) == 4 and all(isinstance(v, int) for v in value)):
            raise TypeError("crop region must be tuple of 4 ints")
        if any(c < 0 for c in value):
            raise ValueError("crop coordinates must be >= 0")
        x0, y0, x1, y1 = value
        self._crop = (min(x0, x1), min(y0, y1), max(x0, x1) + 1, max(y0, y1) + 1)

    @property
    def interpolation(self) -> Interpolation:
        """Filter used when downscaling frames (reference ``scene_manager.py:265-272``)."""
        return self._interpolation

    @interpolation.setter
    def interpolation(self, value: Interpolation):
        self._interpolation = Interpolation(value)

    @property
    def downscale(self) -> int:
        return self._downscale

    @downscale.setter
    def downscale(self, value: int):
        if value < 1:
            raise ValueError("Downscale factor must be a positive integer >= 1!")
        if self.auto_downscale:
            logger.warning("Downscale factor will be ignored because auto_downscale=True!")
        if not isinstance(value, int):
            logger.warning("Downscale factor will be truncated to integer!")
            value = int(value)
        self._downscale = value

    @property
    def auto_downscale(self) -> bool:
        return self._auto_downscale

    @auto_downscale.setter
    def auto_downscale(self, value: bool):
        self._auto_downscale = value

    # -- detectors ---------------------------------------------------------------------------------
    def add_detector(self, detector: SceneDetector) -> None:
        detector.stats_manager = self._stats_manager
        if self._stats_manager is not None:
            self._stats_manager.register_metrics(detector.get_metrics())
        self._detector_list.append(detector)
        self._frame_buffer_size = max(detector.event_buffer_length, self._frame_buffer_size)

    def get_num_detectors(self) -> int:
        return len(self._detector_list)

    def clear(self) -> None:
        self._cutting_list.clear()
        self._last_pos = None
        self._start_pos = None
        self._frame_size = None
        self.clear_detectors()

    def clear_detectors(self) -> None:
        self._detector_list.clear()
        self._carry_frame = None     # (the reference's detectors own this state: scene_manager.py:372-375 drops them)
        self._carry_scale = None

    # -- results -----------------------------------------------------------------------------------
    def _get_cutting_list(self) -> list[FrameTimecode]:
        return sorted(set(self._cutting_list))

    def get_cut_list(self, show_warning: bool = True) -> list[FrameTimecode]:
        """[DEPRECATED in the reference, scene_manager.py:716-744]  The frames at which a new scene starts."""
        if show_warning:
            warnings.warn("get_cut_list() is deprecated and will be removed in a future release.", DeprecationWarning, stacklevel=2)
        return self._get_cutting_list()

    def get_scene_list(self, start_in_scene: bool = False):
        if self._base_timecode is None or self._start_pos is None or self._last_pos is None:
            return []
        cuts = self._get_cutting_list()
        if not cuts and not start_in_scene:
            return []
        return sorted(get_scenes_from_cuts(cuts, self._start_pos, self._last_pos + 1))

    def stop(self) -> None:
        self._stop.set()

    # -- the loop ----------------------------------------------------------------------------------
    def _engine_or_default(self):
        if self._engine is None:
            from pyscenedetect_amd.engine import default_engine

            self._engine = default_engine()
        return self._engine

    def _seen_frame(self, idx: int) -> np.ndarray:
        """The buffered frame as the reference's look-behind buffer holds it (scene_manager.py:422-425: the frame its decode
        thread queued, i.e. cropped and DOWNSCALED).  A call without a callback and without plug-in detectors downloads no
        downscaled frames; should a later call on the same manager hand one of its frames to a callback, it is made here."""
        position, frame, pending = self._frame_buffer[idx]
        if pending is not None:
            frame = self._engine_or_default().downscale_host(np.asarray(frame)[None], pending[0], pending[1])[0]
            self._frame_buffer[idx] = (position, frame, None)
        return frame

    def _dispatch(self, position, frame_im, result, i: int, callback, pending=None) -> bool:
        """One frame's worth of decisions for every detector (reference ``_process_frame`` :410-435).  ``result`` is the
        batch's device output (``_score_batch``), ``i`` the frame's index in it."""
        new_cuts = False
        self._frame_buffer.append((position, frame_im, pending))
        self._frame_buffer = self._frame_buffer[-(self._frame_buffer_size + 1):]
        h, w = result["size"] if result is not None else (frame_im.shape[0], frame_im.shape[1])
        roles = self._roles
        if len(roles) != len(self._detector_list) or any(r[0] is not d for r, d in zip(roles, self._detector_list)):
            roles = self._roles = [(d, _score_flags(d), _thumb_size(d), _edge_kernel(d)) for d in self._detector_list]
        for detector, flags, size, k in roles:      # (what each detector wants from the device: asked once, not per frame)
            if result is not None and result["records"] is not None and flags:
                record = result["records"][i]
                if len(result["edge_xor"]) > 1 and (flags & 8):
                    record = record.copy()      # this detector's own dilation size (content_detector.py:135-137)
                    record["edge_xor"] = result["edge_xor"][k][i]
                cuts = detector.process_record(position, record, h, w)
            elif result is not None and size and size in result["thumbs"]:
                bits = result["bits"].get((size, getattr(detector, "hash_size", None)))
                cuts = detector.process_thumb(position, result["thumbs"][size][i], bits=None if bits is None else bits[i])
            elif isinstance(detector, SceneDetector):
                cuts = detector.process_frame(position, frame_im)
            else:
                # a detector of another library's making (a subclass of the reference's ABC): it is handed the stream's own position
                # object where the stream is of that library too, and what it returns joins the cut list as this package's timecodes
                cuts = [adopt(cut) for cut in detector.process_frame(give_back(position), frame_im)]
            self._cutting_list += cuts
            new_cuts = bool(cuts)
            if callback:
                for cut in cuts:
                    for idx, entry in enumerate(self._frame_buffer):
                        if cut == entry[0]:
                            callback(self._seen_frame(idx), entry[0])
        return new_cuts

    def _plan(self, callback, factor: float) -> dict:
        """What one pass over a batch has to produce for the registered detectors."""
        flags, kernels, thumb_sizes, fallback = 0, [], [], False
        for det in self._detector_list:
            f = _score_flags(det)
            flags |= f
            if f & 8 and _edge_kernel(det) not in kernels:
                kernels.append(_edge_kernel(det))
            if _thumb_size(det) and _thumb_size(det) not in thumb_sizes:
                thumb_sizes.append(_thumb_size(det))
            if not f and not _thumb_size(det):
                fallback = True      # a plug-in detector without a device path: process_frame(frame) like the reference
        # detectors on process_frame() and callbacks get the frame the reference would hand them: the DOWNSCALED one
        # (its decode thread resizes before queueing, scene_manager.py:666-678)
        want_frames = factor > 1.0 and (fallback or callback is not None)
        return {"flags": flags, "kernels": kernels or [0], "thumb_sizes": thumb_sizes, "want_frames": want_frames,
                "device": bool(flags or thumb_sizes), "fallback": fallback}

    def _score_batch(self, engine, plan, frames, factor, last_frame, slot=None, interp: int | None = None):
        """Device results for one batch: records, per-kernel edge counts, thumbnails (+ hash bits), the frames as the
        detectors see them.  ``interp``: the interpolation mode this ``detect_scenes`` call started with (the feeder uploaded
        the rows THAT mode reads; a setter called from a callback takes effect with the next call)."""
        from pyscenedetect_amd import epilogue

        if interp is None:
            interp = self._interpolation.value
        if slot is not None:
            res = engine.analyze_device(slot["ptr"], len(frames), slot["h"], slot["w"], slot["stride"], d_prev=slot["prev"],
                                        flags=plan["flags"], edge_kernels=plan["kernels"], downscale=factor,
                                        hash_sizes=plan["thumb_sizes"], interpolation=interp, want_frames=plan["want_frames"])
        else:
            # engines without device batches (the CPU stand-in of the tests): same results through score_host
            kwargs = {"downscale": factor} if factor > 1.0 else {}
            if factor > 1.0 and interp != 1:
                kwargs["interpolation"] = interp
            stacked = np.stack(frames)
            res = {"records": None, "edge_xor": {}, "thumbs": {}, "frames": None}
            shape = frames[0].shape
            res["size"] = ((max(1, round(shape[0] / factor)), max(1, round(shape[1] / factor))) if factor > 1.0 else shape[:2])
            if plan["flags"]:
                res["records"] = engine.score_host(stacked, prev=last_frame, flags=plan["flags"], edge_kernel=plan["kernels"][0], **kwargs)
                if plan["flags"] & 8:
                    res["edge_xor"][plan["kernels"][0]] = res["records"]["edge_xor"]
                    for k in plan["kernels"][1:]:
                        res["edge_xor"][k] = engine.score_host(stacked, prev=last_frame, flags=8, edge_kernel=k, **kwargs)["edge_xor"]
            for size in plan["thumb_sizes"]:
                res["thumbs"][size] = engine.hash_thumbs_host(stacked, size, **kwargs)
            if plan["want_frames"]:
                res["frames"] = engine.downscale_host(stacked, factor, interp)
        # DCT / median for the whole batch at once (native, threaded) instead of once per frame
        res["bits"] = {}
        for det in self._detector_list:
            size, hs = _thumb_size(det), getattr(det, "hash_size", None)
            if size and hs is not None and size in res["thumbs"] and (size, hs) not in res["bits"]:
                res["bits"][(size, hs)] = epilogue.hash_bits(res["thumbs"][size], hs)
        return res

    def detect_scenes(self, video=None, duration=None, end_time=None, frame_skip: int = 0,
                      show_progress: bool = False,
                      callback: ty.Callable[[np.ndarray, FrameTimecode], None] | None = None, frame_source=None) -> int:
        if frame_source is not None:   # the reference's deprecated spelling of `video` (scene_manager.py:487-494)
            warnings.warn("The `frame_source` argument is deprecated, use `video` instead.", DeprecationWarning, stacklevel=2)
            video = frame_source
        if video is None:
            raise TypeError("detect_scenes() missing 1 required positional argument: 'video'")
        if frame_skip > 0 and self.stats_manager is not None:
            raise ValueError("frame_skip must be 0 when using a StatsManager.")
        if duration is not None and end_time is not None:
            raise ValueError("duration and end_time cannot be set at the same time!")
        if duration is not None and isinstance(duration, (int, float)) and duration < 0:
            raise ValueError("duration must be greater than or equal to 0!")
        if end_time is not None and isinstance(end_time, (int, float)) and end_time < 0:
            raise ValueError("end_time must be greater than or equal to 0!")

        effective_frame_size = video.frame_size
        if self._crop:
            logger.debug(f"Crop set: top left = {self._crop[0:2]}, bottom right = {self._crop[2:4]}")
            x0, y0, x1, y1 = self._crop
            frame_width, frame_height = video.frame_size
            if x0 >= frame_width or y0 >= frame_height:
                raise ValueError("crop starts outside video boundary")
            # (the reference's arithmetic, scene_manager.py:513-525, on the STORED crop, whose far corner is already exclusive:
            #  its warning fires for a crop that ends exactly at the border, and the size the auto-downscale factor is taken from
            #  is one more than the crop's per axis.  Kept as it is -- the factor decides the size of the frames that are scored;
            #  found by tools/fuzz_host_vs_reference.py, round 5.)
            if x1 >= frame_width or y1 >= frame_height:
                logger.warning("Warning: crop ends outside of video boundary.")
            effective_frame_size = (1 + min(x1, frame_width) - x0, 1 + min(y1, frame_height) - y0)
        factor = compute_downscale_factor(max(effective_frame_size)) if self.auto_downscale else self.downscale
        logger.debug("Processing resolution: %d x %d, downscale: %1.1f", int(effective_frame_size[0] / factor),
                     int(effective_frame_size[1] / factor), factor)

        # (every timecode a stream hands over becomes one of this package's: the stream may be one of the reference's backends)
        self._base_timecode = FrameTimecode(video.base_timecode)
        # (the StatsManager is NOT told the base timecode: since round 5 its save_to_csv skips rows keyed by a bare frame number
        #  like the reference's, and a time base planted here would have been state nobody reads)
        start_frame_num = video.frame_number
        if end_time is not None:
            end_time = self._base_timecode + end_time
        elif duration is not None:
            end_time = (self._base_timecode + duration) + start_frame_num

        progress_bar = None
        if show_progress:          # reference scene_manager.py:549-563 (the total from the stream's duration, 0 if unknown)
            total_frames = 0
            if video.duration is not None:
                if end_time is not None and end_time < FrameTimecode(video.duration):
                    total_frames = end_time - start_frame_num
                else:
                    total_frames = video.duration.frame_num - start_frame_num
            progress_bar = _tqdm(total=int(total_frames), unit="frames", desc=PROGRESS_BAR_DESCRIPTION % 0, dynamic_ncols=True)
        prev_position = None

        plan = self._plan(callback, factor)
        self._roles = []                            # (asked again with the first frame of every call: a detector's wishes may have changed)
        interp = self._interpolation.value          # one mode for the whole call: feeder and scoring must agree on it
        # (a manager of plug-in detectors only still needs the engine behind a downscale: the frames they are handed are made there)
        engine = self._engine_or_default() if (plan["device"] or plan["want_frames"]) else None
        # Engines with device batches: the decode thread uploads every frame straight into one of three device batch
        # buffers while this thread scores and decides the previous batch (host -> device copies overlap everything else).
        feeder = _DeviceFeeder(engine, self._batch_frames, factor, interp) if engine is not None and hasattr(engine, "analyze_device") else None

        batches: queue.Queue = queue.Queue(1 if feeder else 2)
        self._stop.clear()
        self._exception_info = None
        worker = threading.Thread(target=self._decode_thread, args=(video, frame_skip, end_time, batches, feeder), daemon=True)
        worker.start()
        # The reference's detectors keep what they derived from the last frame they saw (content_detector.py:189 ...), so a manager
        # that is run on another video WITHOUT clear() scores that video's first frame against the previous video's last one.
        # Here the previous frame belongs to the manager's shared pass: it survives the call, until clear() / clear_detectors()
        # (a detector added in between still starts without a predecessor: each detector has its own "seen a frame" flag).
        last_frame = self._carry_frame
        carry_scale = self._carry_scale if last_frame is not None else None
        # ... and the detectors know better where they have been in the meantime: fed by hand through process_frame(), or run under
        # ANOTHER manager (which left its last frame with them, below) -- the frame a detector saw last is the predecessor then
        if engine is not None:
            for detector in self._detector_list:
                scorer = getattr(detector, "_scorer", None)
                if _score_flags(detector) & 9 and scorer is not None and hasattr(scorer, "last_frame"):
                    seen = scorer.last_frame()
                    if seen is not None:
                        if seen is not last_frame:
                            carry_scale = None      # (a frame fed by hand: as the stream delivered it)
                        last_frame = seen
                        break
        if feeder and last_frame is not None:
            feeder.seed_halo(last_frame)
        logger.info("Detecting scenes...")
        try:
            while not self._stop.is_set():
                batch = batches.get()
                if batch is None:
                    break
                frames, positions, slot = batch
                result = None
                if engine is not None:
                    seam = None
                    if last_frame is not None and (last_frame.shape != frames[0].shape or
                                                   (carry_scale is not None and carry_scale != (factor, interp))):
                        # The frame a detector saw last is of another size, or was scored behind another downscale.  The reference's
                        # ContentDetector compares the planes it KEPT -- downscaled as they were -- with the new frame's behind
                        # `assert left.shape == right.shape` (content_detector.py:29-36): what decides is the size the detectors see.
                        # (Until round 6 the raw shapes were compared: 512 x 288 at factor 2 followed by 256 x 144 at factor 1 raised
                        # here and not in the reference, and a changed `downscale` on frames of one size went unnoticed.)
                        # (no scale on record: the frame was fed BY HAND -- the reference's detector holds it as given, i.e. a caller
                        #  who follows the reference's rules fed frames of the size the detectors see: factor 1 on that side)
                        old_f, old_i = carry_scale if carry_scale is not None else (1.0, interp)
                        seen = _scaled_shape(last_frame.shape, old_f)
                        now = _scaled_shape(frames[0].shape, factor)
                        users = [d for d in self._detector_list if _score_flags(d) & 9 and getattr(d, "_have_last", False)]
                        if seen != now:
                            if users:      # a second video of another size without clear(), a manager taking over from other frames
                                raise AssertionError("frame size changed from %dx%d to %dx%d" % (seen[1], seen[0], now[1], now[0]))
                        elif users:
                            # same size for the detectors, another path to it: the first record of this call is scored on the two small
                            # frames themselves (each behind its own downscale), the batch itself without a predecessor
                            seam = (engine.downscale_host(np.asarray(last_frame)[None], old_f, old_i)[0],
                                    engine.downscale_host(np.asarray(frames[0])[None], factor, interp)[0])
                        last_frame = None
                        if feeder:
                            feeder.forget_halo()
                    carry_scale = None
                    if slot is not None:
                        slot["prev"] = feeder.halo_ptr(slot)
                    result = self._score_batch(engine, plan, frames, factor, last_frame, slot, interp)
                    if seam is not None:
                        _patch_first_record(engine, plan, result, seam)
                    if slot is not None:
                        feeder.release(slot, keep_last=len(frames))
                    last_frame = frames[-1]
                shown = result["frames"] if result is not None and result.get("frames") is not None else frames
                pending = (factor, interp) if factor > 1.0 and shown is frames else None
                for i, position in enumerate(positions):
                    new_cuts = self._dispatch(position, shown[i], result, i, callback, pending)
                    if progress_bar is not None:      # reference scene_manager.py:585-596
                        if new_cuts:
                            progress_bar.set_description(PROGRESS_BAR_DESCRIPTION % len(self._cutting_list), refresh=False)
                        # by the position's delta rather than 1: a VFR stream's frame count is an approximation
                        progress_bar.update(1 if prev_position is None else position.frame_num - prev_position.frame_num)
                        prev_position = position
        finally:
            if progress_bar is not None:
                progress_bar.set_description(PROGRESS_BAR_DESCRIPTION % len(self._cutting_list), refresh=True)
                progress_bar.close()
            self._stop.set()
            if feeder:
                feeder.abort()
            while worker.is_alive():
                while not batches.empty():
                    batches.get_nowait()
                worker.join(timeout=0.1)
            if feeder:
                feeder.close()
        if self._exception_info is not None:
            exc = self._exception_info[1]
            raise exc.with_traceback(self._exception_info[2])
        if last_frame is not None:
            # this manager's (and its detectors' scorers') OWN copy: `frames[-1]` of an ArrayVideoStream is a view of the caller's
            # array -- holding it kept the whole clip alive until clear(), and a stream that reuses its read buffer, or a caller
            # who edits the array between two calls, changed the predecessor under the manager.  One frame per call.
            last_frame = np.array(last_frame, copy=True)
        self._carry_frame = last_frame
        self._carry_scale = (factor, interp) if last_frame is not None else None
        if last_frame is not None:      # (the other way round: process_frame() on one of these detectors goes on from this frame)
            for detector in self._detector_list:
                scorer = getattr(detector, "_scorer", None)
                if scorer is not None and hasattr(scorer, "seed") and _score_flags(detector) & 9:
                    try:
                        scorer.seed(last_frame, scale=(factor, interp))
                    except TypeError:      # (a scorer of the caller's own with the older signature)
                        scorer.seed(last_frame)
        self._last_pos = FrameTimecode(video.position)
        for detector in self._detector_list:
            if isinstance(detector, SceneDetector):
                self._cutting_list += detector.post_process(FrameTimecode(video.position))
            else:
                self._cutting_list += [adopt(cut) for cut in detector.post_process(video.position)]
        return video.frame_number - start_frame_num

    def _decode_thread(self, video, frame_skip: int, end_time, out_queue: queue.Queue, feeder=None) -> None:
        """Reads, size-checks and crops frames and hands them over in batches
        (reference ``_decode_thread`` :625-710; the downscale of :670-678 runs on the device).  With a device feeder
        every frame is copied into the current device batch as soon as it is read."""
        frames: list[np.ndarray] = []
        positions: list[FrameTimecode] = []
        slot = None

        def flush():
            nonlocal frames, positions, slot
            if frames:
                if feeder is not None and slot is not None:
                    feeder.seal(slot)        # the batch's last uploads are enqueued and ordered in front of its scoring
                out_queue.put((frames, positions, slot))
                frames, positions, slot = [], [], None

        try:
            # This thread is the manager's own: it runs on the CPUs of the GPU's NUMA node, so that the frames a decoder allocates in
            # ``video.read()`` are first touched next to the GPU (with the frames on the other socket the same feed moves 24 k instead
            # of 31 k 1080p frames/s).  Nothing to do on one-node hosts or with PSD_FEED_NUMA=0.
            if feeder is not None:
                _run_near_gpu(feeder._engine)
            while not self._stop.is_set():
                frame_im = video.read()
                if frame_im is False:
                    break
                decoded_size = (frame_im.shape[1], frame_im.shape[0])
                if self._frame_size is None:
                    self._frame_size = decoded_size
                    if video.frame_size != decoded_size:
                        logger.warning(f"WARNING: Decoded frame size ({decoded_size}) does not match "
                                       f" video resolution {video.frame_size}, possible corrupt input.")
                elif self._frame_size != decoded_size:
                    self._frame_size_errors += 1
                    if self._frame_size_errors <= MAX_FRAME_SIZE_ERRORS:
                        logger.error(f"ERROR: Frame at {video.position!s} has incorrect size and cannot be "
                                     f"processed: decoded size = {decoded_size}, expected = {self._frame_size}. "
                                     "Video may be corrupt.")
                    if self._frame_size_errors == MAX_FRAME_SIZE_ERRORS:
                        logger.warning("WARNING: Too many errors emitted, skipping future messages.")
                    continue
                if self._crop:
                    x0, y0, x1, y1 = self._crop
                    frame_im = frame_im[y0:y1, x0:x1]
                if self._start_pos is None:
                    self._start_pos = FrameTimecode(video.position)
                if feeder is not None:
                    if slot is not None and slot["shape"] != frame_im.shape:
                        flush()              # a batch holds frames of one size
                    if slot is None:
                        slot = feeder.acquire(frame_im.shape)
                        if slot is None:     # stopped while waiting for a free buffer
                            break
                    feeder.put(slot, len(frames), frame_im)
                frames.append(frame_im)
                position = video.position                        # a copy of this package's kind; the reference's FrameTimecode is
                position = FrameTimecode(position) if isinstance(position, FrameTimecode) else adopt(position)   # adopted, its origin kept
                positions.append(position)
                if len(frames) >= self._batch_frames:
                    flush()
                if frame_skip > 0:
                    for _ in range(frame_skip):
                        if not video.read(decode=False):
                            break
                    position = FrameTimecode(video.position)
                if end_time is not None and not (position + 1) < end_time:
                    break
            flush()
        except KeyboardInterrupt:
            self._stop.set()
        except BaseException:
            logger.critical("Fatal error: Exception raised in decode thread.")
            self._exception_info = sys.exc_info()
            self._stop.set()
        finally:
            if self._start_pos is None:
                self._start_pos = FrameTimecode(video.position)
            out_queue.put(None)


def _run_near_gpu(engine) -> None:
    """Move the CALLING thread onto the CPUs of the GPU's NUMA node (``ScoringEngine.cpus_near_gpu``); best effort."""
    try:
        cpus = engine.cpus_near_gpu() if hasattr(engine, "cpus_near_gpu") else []
        if cpus:
            os.sched_setaffinity(0, cpus)
    except (OSError, AttributeError, RuntimeError):
        pass


class _DeviceFeeder:
    """Three device batch buffers cycling between the decode thread (fills one, frame by frame, with ``psd_upload``) and
    the scoring thread (``SceneManager.detect_scenes``).  The frame preceding a batch lives in a small buffer of its own:
    the last frame of batch k is copied there on the engine's stream before buffer k is handed back (the model is the
    bounded prefetch of the reference's ``_fan_out.py:39-154`` / ``Queue(4)`` of ``scene_manager.py:565-572``, with HBM as
    the queue's storage).  The feeder owns its slots: ``close()`` frees every buffer whoever holds the slot at that moment
    (the queue, the decode thread after an abort, an exception in flight)."""

    N_SLOTS = 3
    FEED_BATCH = 16     # frames per psd_upload_rows_batch call: 16 x 1.66 MB at 1080p -> 256 x 144, about half a millisecond of PCIe
    # Behind a downscale only the source rows that carry taps cross PCIe (engine.TapRowPolicy / psd_upload_rows), into their
    # own places of the full-size device frame -- the rows in between are never read: everything the detectors and callbacks
    # get is computed from the downscaled frame (reference scene_manager.py:666-678).

    def __init__(self, engine, batch_frames: int, factor: float = 1.0, interpolation: int = 1):
        self._engine = engine
        self._batch = batch_frames
        self._factor = factor
        self._interpolation = interpolation
        self._free: queue.Queue = queue.Queue()
        self._all = [{"id": i, "buf": None, "shape": None} for i in range(self.N_SLOTS)]
        self._halo = None          # (DeviceBuffer, shape): the last frame of the most recent batch
        self._have_halo = False
        self._stopped = False
        for slot in self._all:
            self._free.put(slot)

    def acquire(self, shape):
        """A free buffer shaped for frames of ``shape`` (decode thread; waits for the scoring thread if all are busy)."""
        while not self._stopped:
            try:
                slot = self._free.get(timeout=0.1)
            except queue.Empty:
                continue
            h, w, _ = shape
            stride = (h * w * 3 + 15) & ~15
            need = stride * self._batch
            if slot["buf"] is None or slot["buf"].nbytes < need:
                if slot["buf"] is not None:
                    slot["buf"].free()
                    slot["buf"] = None
                slot["buf"] = self._engine.alloc(need)
            slot.update(shape=tuple(shape), h=h, w=w, stride=stride, ptr=slot["buf"].ptr, prev=None, rows=self._rows_of(h, w),
                        pending=[], pending_first=0)
            slot["batched"] = slot["rows"] is not None and hasattr(slot["buf"], "upload_rows_batch") and hasattr(self._engine, "upload_fence")
            return slot
        return None

    def _rows_of(self, h: int, w: int):
        """The rows of an h x w frame that have to be uploaded, or None for all of them."""
        policy = getattr(self._engine, "tap_rows", None)
        return policy(h, w, self._factor, self._interpolation) if policy is not None else None

    def put(self, slot, index: int, frame: np.ndarray) -> None:
        if slot["batched"]:
            # tap rows only, FEED_BATCH frames per call: gathered into page-locked memory by the engine's worker threads and
            # moved by one asynchronous copy (two blocking strided copies per frame reached 60 % of the link)
            if not slot["pending"]:
                slot["pending_first"] = index
            slot["pending"].append(frame)
            if len(slot["pending"]) >= self.FEED_BATCH:
                self._flush_pending(slot)
        elif slot["rows"] is not None:
            slot["buf"].upload_rows(np.ascontiguousarray(frame), index * slot["stride"], slot["rows"])
        else:
            slot["buf"].upload_unordered(np.ascontiguousarray(frame).reshape(-1), index * slot["stride"])

    def _flush_pending(self, slot) -> None:
        if slot["pending"]:
            slot["buf"].upload_rows_batch(slot["pending"], slot["pending_first"] * slot["stride"], slot["rows"], slot["stride"])
            slot["pending"] = []

    def seal(self, slot) -> None:
        """The batch is complete (decode thread): enqueue what is still pending and make the engine's stream wait for the
        copy stream, so that the kernels the scoring thread launches for this batch find every row in place."""
        if slot["batched"]:
            self._flush_pending(slot)
            self._engine.upload_fence()

    def halo_ptr(self, slot):
        """Device address of the frame preceding this batch (None for the first batch or after a size change)."""
        if not self._have_halo or self._halo[1] != slot["shape"]:
            return None
        return self._halo[0].ptr

    def forget_halo(self) -> None:
        self._have_halo = False

    def seed_halo(self, frame: np.ndarray) -> None:
        """The frame that precedes the first batch (the previous call's last frame): uploaded whole, once per detect_scenes()."""
        frame = np.ascontiguousarray(frame)
        h, w, _ = frame.shape
        nbytes = (h * w * 3 + 15) & ~15
        if self._halo is None or self._halo[0].nbytes < nbytes:
            if self._halo is not None:
                self._halo[0].free()
            self._halo = (self._engine.alloc(nbytes), tuple(frame.shape))
        self._halo = (self._halo[0], tuple(frame.shape))
        self._halo[0].upload(frame.reshape(-1))
        self._have_halo = True

    def release(self, slot, keep_last: int) -> None:
        """The batch is scored: keep its last frame for the next batch, hand the buffer back to the decode thread."""
        nbytes = slot["stride"]
        if self._halo is None or self._halo[0].nbytes < nbytes:
            if self._halo is not None:
                self._halo[0].free()
            self._halo = (self._engine.alloc(nbytes), slot["shape"])
        self._halo = (self._halo[0], slot["shape"])
        self._engine.copy_d2d(self._halo[0].ptr, slot["buf"].ptr + (keep_last - 1) * slot["stride"], nbytes)
        # the batch's results were collected before we got here, so this copy is the only work left on the stream that
        # reads the buffer: wait for it, then the decode thread may overwrite the buffer
        self._engine.synchronize()
        self._have_halo = True
        self._free.put(slot)

    def abort(self) -> None:
        self._stopped = True

    def close(self) -> None:
        """Free every buffer (call once the decode thread has stopped)."""
        self._stopped = True
        for slot in self._all:
            if slot["buf"] is not None:
                slot["buf"].free()
                slot["buf"] = None
        if self._halo is not None:
            self._halo[0].free()
            self._halo = None
