"""MI355X-native per-frame scoring engine behind PySceneDetect's detector API.

Drop-in surface (same names and semantics as the reference's ``scenedetect`` package for the hot
path): :class:`SceneManager`, :class:`ContentDetector`, :class:`AdaptiveDetector`,
:class:`HistogramDetector`, :class:`ThresholdDetector`, :class:`HashDetector`, :class:`FrameTimecode`,
:class:`StatsManager`, :class:`SceneDetector`, :class:`FlashFilter`; ``engine`` (the C-ABI binding) and ``epilogue``
(whole-clip native decisions) are the batch face underneath.
"""

from pyscenedetect_amd import corpus, engine, epilogue  # noqa: F401  (submodules of the public surface; the library loads on first use)
from pyscenedetect_amd.detector import FlashFilter, SceneDetector
from pyscenedetect_amd.detectors import AdaptiveDetector, ContentDetector, HashDetector, HistogramDetector, ThresholdDetector
from pyscenedetect_amd.scene_manager import (Interpolation, SceneManager, compute_downscale_factor, expand_scenes_to_bounds,
                                             get_scenes_from_cuts)
from pyscenedetect_amd.stats_manager import FrameMetricRegistered, StatsFileCorrupt, StatsManager
from pyscenedetect_amd.timecode import (CropRegion, CutList, FrameRate, FrameTimecode, SceneList, Timecode, TimecodeLike,
                                        TimecodePair)
from pyscenedetect_amd.video_stream import ArrayVideoStream, FrameRateUnavailable, SeekError, VideoOpenFailure, VideoStream

__version__ = "0.1.0"


def open_video(path, frame_rate=None, backend: str = "opencv", **kwargs):
    """Counterpart of ``scenedetect.open_video()`` (reference ``scenedetect/__init__.py:89-157``).  Decoding is outside this
    package: where the reference is installed its backends do it -- their ``VideoStream`` objects are
    read by this package's ``SceneManager`` unchanged -- and without it this raises ``VideoOpenFailure``."""
    try:
        import scenedetect as _reference
    except ImportError as ex:
        raise VideoOpenFailure("pyscenedetect_amd decodes nothing itself: install scenedetect for its video backends, or pass decoded "
                               "frames (an array uint8[N,H,W,3] or an ArrayVideoStream)") from ex
    if frame_rate is not None:
        kwargs["frame_rate"] = frame_rate
    return _reference.open_video(path, backend=backend, **kwargs)


def detect(video_path, detector, stats_file_path=None, show_progress=False, start_time=None, end_time=None,
           start_in_scene=False, backend: str = "opencv", fps=25.0, engine=None):
    """Counterpart of ``scenedetect.detect()`` (reference ``scenedetect/__init__.py:160-219``, same parameters in the same order;
    ``fps`` and ``engine`` are this package's): ``video_path`` is an array
    ``uint8[N,H,W,3]`` of BGR frames (at ``fps``), a frame source (anything with the ``VideoStream`` members
    ``SceneManager.detect_scenes`` uses, e.g. one of the reference's backends), or -- like the reference's ``video_path`` -- a path
    or list of paths, opened by :func:`open_video` with ``backend``.  Returns the scene list ``[(start, end), ...]``."""
    import os as _os

    import numpy as _np

    video = video_path
    if isinstance(video, _np.ndarray):
        video = ArrayVideoStream(video, fps)
    elif isinstance(video, (str, _os.PathLike)) or (isinstance(video, (list, tuple)) and video
                                                    and all(isinstance(v, (str, _os.PathLike)) for v in video)):
        video = open_video(video, backend=backend)
    if start_time is not None:      # (both times through the constructor first, as in the reference: its checks, its error texts)
        video.seek(FrameTimecode(start_time, video.frame_rate))
    end_timecode = FrameTimecode(end_time, video.frame_rate) if end_time is not None else None
    manager = SceneManager(StatsManager() if stats_file_path else None, engine=engine)
    manager.add_detector(detector)
    manager.detect_scenes(video, end_time=end_timecode, show_progress=show_progress)
    if manager.stats_manager is not None:
        manager.stats_manager.save_to_csv(stats_file_path)
    return manager.get_scene_list(start_in_scene=start_in_scene)

__all__ = [
    "AdaptiveDetector", "ArrayVideoStream", "compute_downscale_factor", "ContentDetector", "CropRegion", "CutList", "detect",
    "expand_scenes_to_bounds", "FlashFilter", "FrameMetricRegistered", "FrameRate", "FrameRateUnavailable", "FrameTimecode", "get_scenes_from_cuts",
    "HashDetector", "HistogramDetector", "Interpolation", "open_video", "SceneDetector", "SceneList", "SceneManager", "SeekError", "StatsFileCorrupt",
    "StatsManager", "ThresholdDetector", "Timecode", "TimecodeLike", "TimecodePair", "VideoOpenFailure", "VideoStream",
]
