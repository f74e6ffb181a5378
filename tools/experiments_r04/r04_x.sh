#!/bin/bash
# round 4, step x: full GPU test suite on the AC=16 build; the fused downscale kernels on U / S / K content (their luma histogram has ONE copy in LDS)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_x; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 --timeout-method=thread > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed|rc=" $O/pytest_gpu.log
for det in all hist content; do for d in U S K; do
  timeout 200 python bench.py --downscale auto --detector $det --dist $d --no-cpu-baseline --no-secondary --steps 10 2>/dev/null | tail -1 > $O/ds_${det}_$d.json
  python -c "import json; d=json.load(open('$O/ds_${det}_$d.json')); print('downscale auto', '$det', '$d', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d.get('parity_sample','')[:50])"
done; done 2>&1 | tee $O/downscale_by_content.txt
