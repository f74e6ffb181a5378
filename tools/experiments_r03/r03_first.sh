#!/bin/bash
# First GPU call of round 3: all GPU tests, the default bench line, the two flow workloads at full length, and HBM traffic
# counters (FETCH_SIZE / WRITE_SIZE, separate passes) for the edge pipeline and the fused all-detectors pass.
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_a}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python - <<PY
import json
try:
    r=json.load(open("$O/bench_default.json"))
    print("headline", r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r.get("parity_sample"))
    for k,v in (r.get("secondary") or {}).items():
        print("  ", k, {kk:vv for kk,vv in v.items() if kk in ("value","avg_launch_ms","frac_of_8TBps","error","parity_sample","ms_per_step")})
        for kk,vv in v.items():
            if isinstance(vv, dict) and "value" in vv: print("      ", kk, {a:b for a,b in vv.items() if a in ("value","avg_launch_ms","frac_of_8TBps","parity_sample","frac")})
    print("  cpu", {k:v for k,v in r["cpu_baseline"].items() if k not in ("sample",)})
except Exception as ex: print("bench parse failed", ex)
PY
timeout 600 python bench.py --workload corpus --steps 5 --warmup 2 > $O/bench_corpus.json 2> $O/bench_corpus.err; echo "corpus rc=$?"; cut -c1-1500 $O/bench_corpus.json
timeout 600 python bench.py --workload bbc --steps 5 --warmup 2 > $O/bench_bbc.json 2> $O/bench_bbc.err; echo "bbc rc=$?"; cut -c1-1500 $O/bench_bbc.json
timeout 300 python bench.py --gpus 1 --dist S --steps 10 --no-secondary --no-cpu-baseline > $O/bench_S_4096.json 2>/dev/null; cut -c1-400 $O/bench_S_4096.json
cd /tmp; export TMPDIR=/tmp
P=$O/pmc_edges; mkdir -p $P
run() { name=$1; shift; ET_N=256 ET_SMOOTH=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/tools/edge_time.py > $P/$name.log 2>&1; }
run f FETCH_SIZE
run w WRITE_SIZE
python $R/tools/pmc_by_kernel.py $P > $O/pmc_edges_traffic.txt; cat $O/pmc_edges_traffic.txt | cut -c1-260
rm -rf $P
P=$O/pmc_fused; mkdir -p $P
run2() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --detector all --frames 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $P/$name.log 2>&1; }
run2 f FETCH_SIZE
run2 w WRITE_SIZE
python $R/tools/pmc_by_kernel.py $P score_frames > $O/pmc_fused_traffic.txt; cat $O/pmc_fused_traffic.txt | cut -c1-260
rm -rf $P
ls $O
