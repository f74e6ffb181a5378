#!/bin/bash
# round 4, step y: four copies of the tile's luma histogram in the fused downscale kernel (tests of the downscaled paths + U / S / K rates)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_z; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 --timeout-method=thread -k "downscale or switch or golden or scene_manager or feed" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=" $O/pytest_gpu.log
for det in all hist; do for d in U S K; do
  timeout 200 python bench.py --downscale auto --detector $det --dist $d --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | tail -1 > $O/ds_${det}_$d.json
  python -c "import json; d=json.load(open('$O/ds_${det}_$d.json')); print('downscale auto', '$det', '$d', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity_sample','')[:60])"
done; done 2>&1 | tee $O/downscale_by_content.txt
