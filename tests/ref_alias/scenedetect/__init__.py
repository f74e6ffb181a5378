"""`scenedetect` -> `pyscenedetect_amd` (see ../README.md)."""
from pyscenedetect_amd import *  # noqa: F401,F403
