"""``scenedetect.common`` by its own name (reference ``scenedetect/common.py``: FrameTimecode, Timecode, Interpolation,
``framerate_to_fraction`` and the type aliases), so that an import written for the reference needs only the package name changed:
``from pyscenedetect_amd.common import FrameTimecode``.  The implementations live in :mod:`pyscenedetect_amd.timecode` and
:mod:`pyscenedetect_amd.scene_manager`."""
from pyscenedetect_amd.scene_manager import Interpolation
from pyscenedetect_amd.timecode import *  # noqa: F401,F403
from pyscenedetect_amd.timecode import (MAX_FPS_DELTA, CropRegion, CutList, FrameRate, FrameTimecode, SceneList, Timecode, TimecodeLike,
                                        TimecodePair, framerate_to_fraction)

__all__ = ["MAX_FPS_DELTA", "CropRegion", "CutList", "FrameRate", "FrameTimecode", "Interpolation", "SceneList", "Timecode", "TimecodeLike",
           "TimecodePair", "framerate_to_fraction"]
