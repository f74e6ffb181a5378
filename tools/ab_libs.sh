#!/bin/bash
# kernel time of `bench.py <args>` for the default library and for builds made by tools/ablate.sh -f NAME
# usage: tools/ab_libs.sh "<bench args>" name1 name2 ...
R=${GRAFT_REPO_ROOT:-$PWD}; ARGS="$1"; shift
t() { PSD_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2 $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-20s' % '$1', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
t default $R/pyscenedetect_amd/libpsd_hip.so
for v in "$@"; do t $v $R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; done
t default $R/pyscenedetect_amd/libpsd_hip.so
