#!/bin/bash
# A/B of ablated builds (tools/ablate.sh -f NAME ...): kernel time of the default bench for each library, then PMC passes.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-r02c}; mkdir -p $O; shift
t() { PSD_LIB_PATH=$2 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % '$1', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
{
t default $R/pyscenedetect_amd/libpsd_hip.so
for v in "$@"; do t $v $R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; done
t default $R/pyscenedetect_amd/libpsd_hip.so
} | tee $O/ablate.txt
