"""MI355X-native per-frame scoring engine behind PySceneDetect's detector API."""

__version__ = "0.1.0"
