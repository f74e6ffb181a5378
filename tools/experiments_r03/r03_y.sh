#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_y}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R

p() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 12 --no-cpu-baseline --no-secondary $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('torchrun %-20s' % '$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
q() { timeout 300 python bench.py --steps 12 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain    %-20s' % '', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" | tee -a $O/ab.txt; }
q; p "--exchange default"; p "--exchange inline"; p "--exchange off"; q; p "--exchange default"; p "--exchange inline"; p "--exchange off"; p "--exchange inline"
