"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): every kernel / copy of the last `--tail-ms`
milliseconds with its start (relative), duration and the idle gap in front of it; totals of busy and idle time."""
import csv
import glob
import sys

d, tail_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
items = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        items.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:70]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        items.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
items.sort()
if not items:
    sys.exit("no trace rows under " + d)
t_end = items[-1][1]
sel = [x for x in items if x[0] >= t_end - tail_ms * 1e6]
t0 = sel[0][0]
busy_end, busy, idle = t0, 0, 0
for s, e, name in sel:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print("%9.1f us  +%8.1f us  gap %7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(gap, 0) / 1e3, name))
    if e > busy_end:
        busy += e - max(busy_end, s)
        busy_end = e
print("window %.3f ms: busy %.3f ms, idle %.3f ms" % ((busy_end - t0) / 1e6, busy / 1e6, idle / 1e6))
