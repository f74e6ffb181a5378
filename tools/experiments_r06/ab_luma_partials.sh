#!/bin/bash
# luma_hist_kernel: tiles leave partial results by plain stores + a reduce kernel (default) against global atomics (PSD_LUMA_PARTIALS=0), interleaved, one box
cd ${GRAFT_REPO_ROOT:-$PWD}
t() { PSD_LUMA_PARTIALS=$1 python bench.py --no-cpu-baseline --no-secondary $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('partials=$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for cfg in "--detector hist --res 4k --frames 2048" "--detector hist --res 4k --frames 2048 --dist S" "--detector hist --frames 4096" "--detector hist --dist K --frames 2048"; do echo "## $cfg"; for i in 1 2 3; do t 1 "$cfg"; t 0 "$cfg"; done; done
