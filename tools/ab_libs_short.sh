#!/bin/bash
# like tools/ab_libs_long.sh at the bench defaults (20 timed steps after 3 warm-up steps): what a default bench line of the variant reads
R=${GRAFT_REPO_ROOT:-$PWD}; ARGS="$1"; ROUNDS=$2; shift 2
t() { if [ "$1" = default ]; then L=$R/pyscenedetect_amd/libpsd_hip.so; else L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$1.so; fi
  PSD_LIB_PATH=$L python bench.py --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s' % '$1', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['value'])"; }
for i in $(seq $ROUNDS); do for v in "$@"; do t $v; done; done
