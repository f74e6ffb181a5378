#!/bin/bash
# round 5, sixth probe: A/B of the clip-start flag read in front of the DMA issue (new) against the build before it
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
OLD=$R/pyscenedetect_amd/csrc/build/abl/libpsd_before_segfix.so
line() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], (d.get('parity_sample') or '')[:30])"; }
run() {  # tag, bench args...
  tag=$1; shift
  for lib in new old new old; do
    [ $lib = old ] && export PSD_LIB_PATH=$OLD || unset PSD_LIB_PATH
    timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/${tag}_$lib.json 2>/dev/null; line $O/${tag}_$lib.json "$tag $lib"
  done
}
run headline --steps 12
run S2048 --dist S --frames 2048 --steps 12
run K2048 --dist K --frames 2048 --steps 12
run content4k --res 4k --frames 2048 --steps 8
run all --detector all --steps 8
run allS --detector all --dist S --steps 8
run edgesS --detector edges --dist S --frames 2048 --steps 4 --warmup 2
run edgesT --detector edges --dist T --frames 2048 --steps 4 --warmup 2
run bbc_small --workload bbc --bbc-frames 2000 --steps 6 --warmup 2
run corpus_small --workload corpus --corpus-frames 512 --steps 6 --warmup 2
run bbc_full --workload bbc --steps 5 --warmup 2
unset PSD_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flows.py tests/test_gpu_headline_geometry.py -m gpu -x -q 2>&1 | tail -3
