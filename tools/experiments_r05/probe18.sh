#!/bin/bash
# round 5, probe 18: completion of small submissions by polling (PSD_SPIN_US, default 400) against the runtime's wait (0), per-frame lines
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export PYTHONPATH=$R:$R/tools
pf() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print('$1 per_frame_us', s['per_frame_api_1080p']['us_per_frame'], 'binding_us', s['per_frame_api_1080p']['reference_binding']['us_per_frame'], 'host_fed', s['host_fed_default_pipeline']['value'], 'headline', d['value'], d['roofline']['frac'], 'bbc', s['bbc_standin_adaptive']['value'], 'corpus', s['corpus_mixed_1080p_4k_all_four']['value'])"; }
for k in 400 0 400 0; do
  export PSD_SPIN_US=$k
  pf spin$k
  timeout 120 python tools/experiments_r05/per_frame_breakdown.py 2>/dev/null | tail -1
done
