from pyscenedetect_amd.scene_manager import *  # noqa: F401,F403
from pyscenedetect_amd.scene_manager import SceneManager  # noqa: F401
