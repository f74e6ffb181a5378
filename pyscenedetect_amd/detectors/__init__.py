"""GPU-backed drop-ins for ``scenedetect.detectors`` (reference ``detectors/__init__.py:38-42``)."""

from pyscenedetect_amd.detectors.content_detector import ContentDetector
from pyscenedetect_amd.detectors.adaptive_detector import AdaptiveDetector
from pyscenedetect_amd.detectors.hash_detector import HashDetector
from pyscenedetect_amd.detectors.histogram_detector import HistogramDetector
from pyscenedetect_amd.detectors.threshold_detector import ThresholdDetector

__all__ = ["ContentDetector", "AdaptiveDetector", "HashDetector", "HistogramDetector", "ThresholdDetector"]
