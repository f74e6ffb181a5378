#!/bin/bash
# round 5, eighth probe: old (round-4 flag read) vs tmpl (SEG instances) vs final (+ hidden SAD adds), all workloads; parity tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
line() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], (d.get('parity_sample') or '')[:30])"; }
run() {  # tag, bench args...
  tag=$1; shift
  for lib in final tmpl old final tmpl old; do
    case $lib in old) export PSD_LIB_PATH=$A/libpsd_before_segfix.so;; tmpl) export PSD_LIB_PATH=$A/libpsd_tmpl.so;; *) unset PSD_LIB_PATH;; esac
    timeout 300 python bench.py --no-cpu-baseline --no-secondary "$@" > $O/${tag}_$lib.json 2>/dev/null; line $O/${tag}_$lib.json "$tag $lib"
  done
}
run headline --steps 12
run S2048 --dist S --frames 2048 --steps 12
run content4k --res 4k --frames 2048 --steps 8
run all --detector all --steps 8
run allS --detector all --dist S --steps 8
run allK --detector all --dist K --frames 2048 --steps 8
run edgesS --detector edges --dist S --frames 2048 --steps 4 --warmup 2
run edgesT --detector edges --dist T --frames 2048 --steps 4 --warmup 2
run bbc_small --workload bbc --bbc-frames 2000 --steps 6 --warmup 2
run corpus_small --workload corpus --corpus-frames 512 --steps 6 --warmup 2
unset PSD_LIB_PATH
