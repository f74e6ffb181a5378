class VideoStreamCv2:
    """Name only: decoding is out of scope; the reference tests that open a video are deselected."""

    def __init__(self, *a, **k):
        raise NotImplementedError("no decoder in this image")
