"""MI355X-native per-frame scoring engine behind PySceneDetect's detector API.

Drop-in surface (same names and semantics as the reference's ``scenedetect`` package for the hot
path): :class:`SceneManager`, :class:`ContentDetector`, :class:`AdaptiveDetector`,
:class:`HistogramDetector`, :class:`ThresholdDetector`, :class:`FrameTimecode`,
:class:`StatsManager`, :class:`SceneDetector`, :class:`FlashFilter`.
"""

from pyscenedetect_amd.detector import FlashFilter, SceneDetector
from pyscenedetect_amd.detectors import AdaptiveDetector, ContentDetector, HistogramDetector, ThresholdDetector
from pyscenedetect_amd.scene_manager import SceneManager, compute_downscale_factor, get_scenes_from_cuts
from pyscenedetect_amd.stats_manager import StatsManager
from pyscenedetect_amd.timecode import FrameTimecode
from pyscenedetect_amd.video_stream import ArrayVideoStream

__version__ = "0.1.0"

__all__ = [
    "AdaptiveDetector", "ArrayVideoStream", "ContentDetector", "FlashFilter", "FrameTimecode", "HistogramDetector",
    "SceneDetector", "SceneManager", "StatsManager", "ThresholdDetector", "compute_downscale_factor",
    "get_scenes_from_cuts",
]
