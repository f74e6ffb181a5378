#!/bin/bash
# luma_hist_kernel's flush: one thread per bin, two bins per 64-bit atomic (default) against one 32-bit atomic per bin in three passes (pair0), interleaved, one box
cd ${GRAFT_REPO_ROOT:-$PWD}
A=pyscenedetect_amd/csrc/build/abl
t() { PSD_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-secondary $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % '$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
run() { echo "## $1"; for i in 1 2 3; do t pair $PWD/pyscenedetect_amd/libpsd_hip.so "$2"; t single $PWD/$A/libpsd_pair0.so "$2"; done; }
run "Histogram + Threshold 4K, U" "--detector hist --res 4k --frames 2048"
run "Histogram + Threshold 4K, S" "--detector hist --res 4k --frames 2048 --dist S"
run "Histogram + Threshold 1080p, U" "--detector hist --frames 4096"
run "Histogram + Threshold 1080p, K" "--detector hist --dist K --frames 2048"
