#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_w; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for i in 1 2; do for v in 0 1; do for d in S T; do
  PSD_EDGE_FUSE_DOWNSCALE=$v python bench.py --downscale auto --detector edges --dist $d --frames 4096 --no-secondary --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused $v $d', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'], str(d.get('parity'))[:80])"
done; done; done 2>&1 | tee $O/ab_edge_fuse_downscale.txt
