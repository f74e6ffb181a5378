#!/bin/bash
# final check of the round's last build: smoke() + the default bench line (with `secondary`, incl. host_fed_default_pipeline)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_au}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
SECONDS=0; timeout 500 python bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS}s"; tail -2 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
for k,v in d.get("secondary",{}).items():
    print(k, {kk:v.get(kk) for kk in ("value","frac_of_8TBps","error","parity_sample","pcie_inclusive","host_to_device_GBps")})
PY
