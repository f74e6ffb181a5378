#!/bin/bash
# round 5, third probe: the flows with a tail piece (A/B over PSD_CLIPS_TAIL_MB), the new GPU tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_flows.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2; do
for mb in 0 default 3072; do
  for w in bbc corpus; do
    if [ $mb = default ]; then unset PSD_CLIPS_TAIL_MB; else export PSD_CLIPS_TAIL_MB=$mb; fi
    timeout 300 python bench.py --workload $w --bbc-frames 2000 --corpus-frames 512 --steps 6 --warmup 2 --cpu-sample 64 > $O/flow_small_${w}_${mb}_$rep.json 2> $O/flow_small_${w}_$mb.err
    python -c "import json; d=json.load(open('$O/flow_small_${w}_${mb}_$rep.json')); print('small $w tail=$mb', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_sample'][:40])"
  done
done
done
for mb in 0 default; do
  for w in bbc corpus; do
    if [ $mb = default ]; then unset PSD_CLIPS_TAIL_MB; else export PSD_CLIPS_TAIL_MB=$mb; fi
    timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 64 > $O/flow_full_${w}_$mb.json 2> $O/flow_full_${w}_$mb.err
    python -c "import json; d=json.load(open('$O/flow_full_${w}_$mb.json')); print('full $w tail=$mb', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_sample'][:40])"
  done
done
