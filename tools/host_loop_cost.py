"""Pure host-loop cost of SceneManager.detect_scenes: a stub engine that moves and computes nothing."""
import sys, time, cProfile, pstats
sys.path[:0] = [__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))]
import numpy as np
import pyscenedetect_amd as psd
from pyscenedetect_amd import _native
from pyscenedetect_amd.engine import RECORD_DTYPE

class Buf:
    def __init__(s, n): s.nbytes = n; s.ptr = 4096
    def upload(s, *a, **k): pass
    upload_unordered = upload_rows = upload_rows_batch = upload
    def free(s): pass
class Stub:
    def alloc(s, n): return Buf(n)
    def tap_rows(s, h, w, factor, interp): return np.arange(0, h, 4, dtype=np.int32) if factor > 1 else None
    def upload_fence(s): pass
    def synchronize(s): pass
    def copy_d2d(s, *a): pass
    def cpus_near_gpu(s): return []
    def analyze_device(s, ptr, n, h, w, stride, d_prev=None, flags=0, edge_kernels=(0,), downscale=1.0, hash_sizes=(), interpolation=1, want_frames=False):
        rec = np.zeros(n, RECORD_DTYPE); rec['sad_v'] = 1000
        fh, fw = (max(1, round(h / downscale)), max(1, round(w / downscale))) if downscale > 1 else (h, w)
        return {"records": rec, "edge_xor": {}, "thumbs": {}, "frames": None, "size": (fh, fw)}
class Frames:
    def __init__(s, n, h, w): s.f = np.zeros((h, w, 3), np.uint8); s.n = n; s.shape = (n, h, w, 3)
    def __len__(s): return s.n
    def __getitem__(s, i): return s.f
def run(n, stats, det='content'):
    e = Stub()
    sm = psd.SceneManager(psd.StatsManager() if stats else None, engine=e)
    sm.add_detector({'content': psd.ContentDetector, 'adaptive': psd.AdaptiveDetector}[det](engine=e))
    v = psd.ArrayVideoStream(Frames(n, 1080, 1920), 25.0)
    t = time.perf_counter(); sm.detect_scenes(v); dt = time.perf_counter() - t
    return dt / n * 1e6
if __name__ == '__main__':
    n = 30000
    for stats in (False, True):
        for det in ('content', 'adaptive'):
            print(f"stats={stats} {det}: {min(run(n, stats, det) for _ in range(7)):.2f} us/frame")
    if len(sys.argv) > 1:
        cProfile.run("run(60000, False)", "/tmp/psd_host_loop.prof")
        pstats.Stats("/tmp/psd_host_loop.prof").sort_stats("tottime").print_stats(18)
