from pyscenedetect_amd import stats_manager as _s

globals().update({k: getattr(_s, k) for k in dir(_s) if not k.startswith("__")})
