"""MI355X-native per-frame scoring engine behind PySceneDetect's detector API.

Drop-in surface (same names and semantics as the reference's ``scenedetect`` package for the hot
path): :class:`SceneManager`, :class:`ContentDetector`, :class:`AdaptiveDetector`,
:class:`HistogramDetector`, :class:`ThresholdDetector`, :class:`HashDetector`, :class:`FrameTimecode`,
:class:`StatsManager`, :class:`SceneDetector`, :class:`FlashFilter`; ``engine`` (the C-ABI binding) and ``epilogue``
(whole-clip native decisions) are the batch face underneath.
"""

from pyscenedetect_amd import engine, epilogue  # noqa: F401  (submodules of the public surface; the library loads on first use)
from pyscenedetect_amd.detector import FlashFilter, SceneDetector
from pyscenedetect_amd.detectors import AdaptiveDetector, ContentDetector, HashDetector, HistogramDetector, ThresholdDetector
from pyscenedetect_amd.scene_manager import (Interpolation, SceneManager, compute_downscale_factor, expand_scenes_to_bounds,
                                             get_scenes_from_cuts)
from pyscenedetect_amd.stats_manager import FrameMetricRegistered, StatsFileCorrupt, StatsManager
from pyscenedetect_amd.timecode import (CropRegion, CutList, FrameRate, FrameTimecode, SceneList, Timecode, TimecodeLike,
                                        TimecodePair)
from pyscenedetect_amd.video_stream import ArrayVideoStream, FrameRateUnavailable, SeekError, VideoOpenFailure, VideoStream

__version__ = "0.1.0"


def detect(video, detector, stats_file_path=None, show_progress=False, start_time=None, end_time=None,
           start_in_scene=False, fps=25.0, engine=None):
    """Counterpart of ``scenedetect.detect()`` (reference ``scenedetect/__init__.py:160-219``) for
    already-decoded frames: ``video`` is a frame source (anything with the ``VideoStream`` members
    ``SceneManager.detect_scenes`` uses, e.g. one of the reference's backends) or an array
    ``uint8[N,H,W,3]`` of BGR frames.  Returns the scene list ``[(start, end), ...]``."""
    import numpy as _np

    if isinstance(video, _np.ndarray):
        video = ArrayVideoStream(video, fps)
    if start_time is not None:
        video.seek(start_time if isinstance(start_time, (int, FrameTimecode)) else FrameTimecode(start_time, video.frame_rate))
    manager = SceneManager(StatsManager() if stats_file_path else None, engine=engine)
    manager.add_detector(detector)
    manager.detect_scenes(video, end_time=end_time, show_progress=show_progress)
    if manager.stats_manager is not None:
        manager.stats_manager.save_to_csv(stats_file_path)
    return manager.get_scene_list(start_in_scene=start_in_scene)

__all__ = [
    "AdaptiveDetector", "ArrayVideoStream", "compute_downscale_factor", "ContentDetector", "CropRegion", "CutList", "detect",
    "expand_scenes_to_bounds", "FlashFilter", "FrameMetricRegistered", "FrameRate", "FrameRateUnavailable", "FrameTimecode", "get_scenes_from_cuts",
    "HashDetector", "HistogramDetector", "Interpolation", "SceneDetector", "SceneList", "SceneManager", "SeekError", "StatsFileCorrupt",
    "StatsManager", "ThresholdDetector", "Timecode", "TimecodeLike", "TimecodePair", "VideoOpenFailure", "VideoStream",
]
