#!/bin/bash
# round 4, step ae: busy Sobel tiles without the per-pixel test in front (the test is part of the result); the default bench once
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_ae; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
B=$R/pyscenedetect_amd/csrc/build/abl/libpsd_base.so
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_headline_geometry.py tests/test_gpu_parity.py tests/test_gpu_switches.py -m gpu -q -x --timeout=600 --timeout-method=thread -k "edge or hysteresis or dilation or serpentine or one_read or switch" > $O/pytest_edges.log 2>&1; echo "pytest rc=$?" >> $O/pytest_edges.log; tail -3 $O/pytest_edges.log
{ PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STU base; python tools/edge_ab.py 2048 STU new; PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STU base; python tools/edge_ab.py 2048 STU new; } 2>&1 | grep -v amdgpu.ids | tee $O/edge_ab.txt
SECONDS=0; timeout 600 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json; echo "bench rc=${PIPESTATUS[0]}"; echo "bench wall seconds: $SECONDS"
python - <<PY
import json
r=json.load(open("$O/bench_default.json"))
print(r["value"], r["roofline"]["frac"], r["parity_sample"][:100])
for k,v in (r.get("secondary") or {}).items():
    print("  ", k, {kk:vv for kk,vv in v.items() if kk in ("value","avg_launch_ms","frac_of_8TBps","error")}, (v.get("parity_sample") or "")[:50])
    for kk,vv in v.items():
        if isinstance(vv, dict) and "value" in vv: print("      ", kk, {a:b for a,b in vv.items() if a in ("value","avg_launch_ms","frac_of_8TBps","error")}, (vv.get("parity_sample") or "")[:50])
PY
