#!/bin/bash
# round 5, probe 17: bench.py's per-frame line reads 170 us on a fresh box and 253 us in the measurement matrices (after the GPU tests).
# Box state?  bench -> GPU tests -> bench -> breakdown tool -> drop the page cache -> bench, with memory state in between.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; export PYTHONPATH=$R:$R/tools
pf() { python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print('$1 per_frame_us', s['per_frame_api_1080p']['us_per_frame'], 'binding_us', s['per_frame_api_1080p']['reference_binding']['us_per_frame'], 'host_fed', s['host_fed_default_pipeline']['value'])"; }
mem() { echo "$1: $(grep -E 'MemFree|^Cached|AnonHugePages|HugePages_Free' /proc/meminfo | tr -s ' ' | tr '\n' ';') thp_alloc=$(grep -E 'thp_fault_alloc |thp_fault_fallback ' /proc/vmstat | tr '\n' ' ')"; numactl -H 2>/dev/null | grep free; }
mem start; pf fresh
timeout 900 python -m pytest tests -m gpu -q -x > /dev/null 2>&1; echo "pytest rc=$?"
mem after_pytest; pf after_pytest
timeout 120 python tools/experiments_r05/per_frame_breakdown.py 2>/dev/null | tail -1
sync; echo 3 > /proc/sys/vm/drop_caches 2>/dev/null && echo dropped
mem after_drop; pf after_drop
