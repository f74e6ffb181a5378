from pyscenedetect_amd import timecode as _t

globals().update({k: getattr(_t, k) for k in dir(_t) if not k.startswith("__")})
