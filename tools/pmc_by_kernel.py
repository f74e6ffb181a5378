"""Per-kernel averages of PMC counters from rocprofv3 --pmc csv output (one sub-directory per counter pass).
usage: python tools/pmc_by_kernel.py <dir-with-pass-subdirs> [substring ...]   (kernels whose name holds any substring)

FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts a wide coalesced read at half its
bytes (MI355X_MICROARCH.md, HBM section), so HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
subs = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", row["Kernel_Name"])
        name = re.sub(r"^void ", "", name)
        if subs and not any(s in name for s in subs):
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    c = acc[name]
    line = [f"{name[:90]:90s}"]
    for k in sorted(c):
        line.append(f"{k}={sum(c[k]) / len(c[k]):.6g} (n={len(c[k])})")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f_, w_ = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]), sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        line.append(f"HBM_bytes_per_launch={(2 * f_ + w_) * 1024:.6g}")
    print("  ".join(line))
