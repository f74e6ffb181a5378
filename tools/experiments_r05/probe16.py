"""round 5, probe 16: is the bimodal per-frame line of bench.py (170 us on some boxes, 253 us on others) the NUMA node the process runs on?
The same process_frame loop with the main thread (and the frames it then allocates) on each node of the host in turn."""
import os, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import pyscenedetect_amd as psd
from pyscenedetect_amd import engine as E
from pyscenedetect_amd.timecode import FrameTimecode
def cpulist(s):
    out = set()
    for part in s.strip().split(","):
        a, _, b = part.partition("-"); out |= set(range(int(a), int(b or a) + 1))
    return out
nodes = {int(p.rsplit("node", 1)[1].split("/")[0]): cpulist(open(p).read()) for p in glob.glob("/sys/devices/system/node/node*/cpulist")}
dev = torch.device("cuda", 0)
eng = E.ScoringEngine(0)
bdf = torch.cuda.get_device_properties(0)
try:
    import ctypes
    pci = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
except Exception: pci = None
gpu_node = None
for p in glob.glob("/sys/bus/pci/devices/*/numa_node"):
    pass
N, H, W = 96, 1080, 1920
b = bench.make_batch(N, "S", 20250921, dev, H, W)
tcs = [FrameTimecode(i, 25.0) for i in range(N)]
def loop(frames):
    det = psd.ContentDetector(engine=eng)
    for i in range(len(frames)): det.process_frame(tcs[i], frames[i])
all_cpus = os.sched_getaffinity(0)
res = {"cpu_now": None, "nodes": {k: len(v) for k, v in nodes.items()}}
for node, cpus in sorted(nodes.items()):
    os.sched_setaffinity(0, cpus & all_cpus)
    time.sleep(0.05)
    frames = np.empty((N, H, W, 3), np.uint8); torch.from_numpy(frames).copy_(b)   # allocated and first touched on this node
    loop(frames[:8]); t = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); loop(frames); t = min(t, time.perf_counter() - t0)
    res[f"node{node}_us_per_frame"] = round(t / N * 1e6, 1)
    # thread here, frames of the OTHER placement: keep the first node's frames for a cross test
    if node == min(nodes): first_frames = frames
    else:
        t = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); loop(first_frames); t = min(t, time.perf_counter() - t0)
        res[f"thread_node{node}_frames_node{min(nodes)}_us"] = round(t / N * 1e6, 1)
os.sched_setaffinity(0, all_cpus)
if hasattr(eng, "near_gpu_cpus"): res["near"] = len(eng.near_gpu_cpus())
print(res)
