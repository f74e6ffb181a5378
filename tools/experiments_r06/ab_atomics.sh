#!/bin/bash
# what the global atomics of the score kernels' flushes cost: ablated builds (WRONG results by design) without the histogram atomics (64), without the
# sums' atomics (128), without both (192); interleaved with the default build, one box
cd ${GRAFT_REPO_ROOT:-$PWD}
A=pyscenedetect_amd/csrc/build/abl
t() { PSD_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-secondary $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % '$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
run() { echo "## $1"; shift; cfg="$1"; shift; for i in 1 2; do t default $PWD/pyscenedetect_amd/libpsd_hip.so "$cfg"; for v in "$@"; do t abl$v $PWD/$A/libpsd_abl$v.so "$cfg"; done; done; }
run "headline" "" 128
run "all four fused, full resolution" "--detector all" 64 192
run "Histogram + Threshold 4K" "--detector hist --res 4k --frames 2048" 64 192
run "edge term + HSV, S" "--detector edges --dist S --frames 2048 --steps 5 --warmup 2" 64 192
