#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_z}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench.json 2>$O/err.log
ls $O/trace | head; head -30 $O/trace/t_kernel_stats.csv | cut -c1-200
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/trace/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# print a window of consecutive kernels around the 5th HSV launch
idx=[i for i,r in enumerate(rows) if "score_frames_dma" in r["Kernel_Name"]]
i0=idx[5]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0-1:idx[7]+1]:
    print("%10.1f us  +%9.1f us  q=%s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id"), r["Kernel_Name"][:90]))
PY
rm -rf $O/trace
