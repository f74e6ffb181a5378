from pyscenedetect_amd.detector import FlashFilter, SceneDetector  # noqa: F401
