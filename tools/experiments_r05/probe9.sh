#!/bin/bash
# round 5, ninth probe: LDS-only barriers (hash DMA kernel, time walks, downscale walk) against the build before (final1); GPU tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
line() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], (d.get('parity_sample') or '')[:30])"; }
run() {  # tag, bench args...
  tag=$1; shift
  for lib in new final1 new final1; do
    case $lib in final1) export PSD_LIB_PATH=$A/libpsd_final1.so;; *) unset PSD_LIB_PATH;; esac
    timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/${tag}_$lib.json 2>/dev/null; line $O/${tag}_$lib.json "$tag $lib"
  done
}
run hash --detector hash --steps 6
run hash4k --detector hash --res 4k --frames 1024 --steps 6
run headline --steps 12
run all --detector all --steps 8
run allS --detector all --dist S --steps 8
run edgesS --detector edges --dist S --frames 2048 --steps 4 --warmup 2
run downscale --downscale auto --steps 20
run downscale_all --downscale auto --detector all --steps 20
run corpus_small --workload corpus --corpus-frames 512 --steps 6 --warmup 2
unset PSD_LIB_PATH
