#!/bin/bash
# round 4, GPU call b: whole GPU suite + the default bench line (with the new parity samples and per-frame lines)
out=gpurun_out/r04_b; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
timeout 1200 python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
tail -15 $out/pytest_gpu.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"
tail -3 $out/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_b/bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["parity_sample"])
for k,v in d["secondary"].items():
    print(k, {kk:vv for kk,vv in v.items() if kk in ("value","frac_of_8TBps","parity_sample","error","ms_per_step")})
    for kk,vv in v.items():
        if isinstance(vv,dict) and "value" in vv: print("   ",kk, vv.get("value"), vv.get("frac_of_8TBps"), vv.get("parity_sample"))
PY
