"""Average PMC counter values per dispatch of a kernel from rocprofv3 csv output.
usage: python tools/pmc_summary.py <dir-with-pass-subdirs> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "score_frames_kernel"
for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if want in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        print(f"{os.path.basename(os.path.dirname(f)):6s} {k:28s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
