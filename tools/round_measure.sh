#!/bin/bash
# End-of-round measurement matrix on one MI355X.  usage: tools/round_measure.sh <tag>   (writes gpurun_out/<tag>_*)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-rXX}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout=300 --timeout-method=thread > $O/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> $O/${T}_pytest.log; tail -2 $O/${T}_pytest.log
timeout 200 python __graft_entry__.py smoke > $O/${T}_smoke.log 2>&1; tail -1 $O/${T}_smoke.log
b() { name=$1; shift; timeout 300 python bench.py "$@" 2>/dev/null | tail -1 > $O/${T}_bench_$name.json; python -c "
import json;d=json.load(open('$O/${T}_bench_$name.json'));print('$name',d['value'],d['roofline']['frac'],d['ms_per_step'],(d.get('cpu_baseline') or {}).get('value'))"; }
b content
b content_S --no-cpu-baseline --dist S
b content_4k --no-cpu-baseline --res 4k --frames 1024
b hist --no-cpu-baseline --detector hist
b hist_4k --no-cpu-baseline --detector hist --res 4k --frames 1024
b all --no-cpu-baseline --detector all
timeout 200 python tools/hash_time.py 1024 > $O/${T}_hash_thumbs.jsonl 2>/dev/null; cat $O/${T}_hash_thumbs.jsonl
for n in 64 1024; do ET_N=$n ET_SMOOTH=1 timeout 120 python tools/edge_time.py 2>&1 | tail -1; done | tee $O/${T}_edges.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_trace -o t -- python $R/bench.py --no-cpu-baseline > $O/${T}_trace_bench.json 2>/dev/null
cd $R; python tools/rocpd_summary.py $O/${T}_trace/*.db > $O/${T}_kernel_trace.md 2>&1; grep -v "at::native\|rocclr" $O/${T}_kernel_trace.md | cut -c1-220 | head -8; tail -1 $O/${T}_trace_bench.json | cut -c1-400
