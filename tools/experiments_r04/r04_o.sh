#!/bin/bash
# round 4, step o: the edge term on two lanes (tests + A/B inside one process), the launcher line with ONE all-gather per run
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_o; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 900 python tools/edge_lanes_ab.py 2048 STU > $O/edge_lanes_ab.txt 2>&1; cat $O/edge_lanes_ab.txt | grep -v amdgpu.ids
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_torchrun_1rank.json
timeout 300 python bench.py --steps 10 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_plain.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 10 --no-secondary --no-cpu-baseline --exchange step 2>/dev/null | tail -1 > $O/bench_torchrun_1rank_step.json
for f in torchrun_1rank plain torchrun_1rank_step; do python -c "import json; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config'].get('exchange'))"; done
