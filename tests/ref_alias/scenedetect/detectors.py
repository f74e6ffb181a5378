from pyscenedetect_amd.detectors import *  # noqa: F401,F403
from pyscenedetect_amd import AdaptiveDetector, ContentDetector, HashDetector, HistogramDetector, ThresholdDetector  # noqa: F401
