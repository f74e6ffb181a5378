#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for i in 1 2; do for v in 0 1; do
  PSD_RESIZE_STORE_VEC=$v python bench.py --downscale auto --detector edges --dist T --frames 4096 --no-secondary --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('store_vec $v', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'], d.get('parity'))"
done; done 2>&1 | tee $O/ab_store_vec.txt
python tools/downscale_split.py 2>&1 | tail -3
