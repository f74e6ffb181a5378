#!/bin/bash
# round 5, seventh probe: fused passes with / without the DMA drain behind the issue, against the build before the flag fix
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
line() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], (d.get('parity_sample') or '')[:30])"; }
run() {  # tag, bench args...
  tag=$1; shift
  for lib in drain nodrain old drain nodrain old; do
    case $lib in old) export PSD_LIB_PATH=$A/libpsd_before_segfix.so;; nodrain) export PSD_LIB_PATH=$A/libpsd_segfix_nodrain.so;; *) unset PSD_LIB_PATH;; esac
    timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/${tag}_$lib.json 2>/dev/null; line $O/${tag}_$lib.json "$tag $lib"
  done
}
run all --detector all --steps 8
run allS --detector all --dist S --steps 8
run allK --detector all --dist K --frames 2048 --steps 8
run edgesS --detector edges --dist S --frames 2048 --steps 4 --warmup 2
run edgesT --detector edges --dist T --frames 2048 --steps 4 --warmup 2
run corpus_small --workload corpus --corpus-frames 512 --steps 6 --warmup 2
run corpus_full --workload corpus --steps 4 --warmup 2
unset PSD_LIB_PATH
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flows.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
