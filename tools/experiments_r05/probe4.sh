#!/bin/bash
# round 5, fourth probe: launch geometry (workgroups per launch / walk length) of the HSV pass and the fused pass by workload
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
line() { python -c "import json,sys; d=json.load(open('$1')); print('$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
export PSD_CLIPS_TAIL_MB=0
for rep in 1 2; do
for b in 0 16384 65536 131072; do
  [ $b = 0 ] && unset PSD_HSV_BLOCKS || export PSD_HSV_BLOCKS=$b
  timeout 300 python bench.py --workload bbc --steps 5 --warmup 2 --no-cpu-baseline > $O/bbc_full_$b.json 2>/dev/null; line $O/bbc_full_$b.json "bbc full blocks=$b"
  timeout 300 python bench.py --workload bbc --bbc-frames 2000 --steps 6 --warmup 2 --no-cpu-baseline > $O/bbc_small_$b.json 2>/dev/null; line $O/bbc_small_$b.json "bbc small blocks=$b"
done
done
unset PSD_CLIPS_TAIL_MB
for b in 0 24576 49152 65536; do
  [ $b = 0 ] && unset PSD_HSV_BLOCKS || export PSD_HSV_BLOCKS=$b
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 12 > $O/headline_$b.json 2>/dev/null; line $O/headline_$b.json "headline blocks=$b"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --res 4k --frames 2048 --steps 8 > $O/content4k_$b.json 2>/dev/null; line $O/content4k_$b.json "content 4k blocks=$b"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --dist S --frames 2048 --steps 12 > $O/s2048_$b.json 2>/dev/null; line $O/s2048_$b.json "S 2048 blocks=$b"
done
unset PSD_HSV_BLOCKS
for b in 0 1024 4096 8192; do
  [ $b = 0 ] && unset PSD_FUSED_BLOCKS || export PSD_FUSED_BLOCKS=$b
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --detector all --steps 8 > $O/all_$b.json 2>/dev/null; line $O/all_$b.json "all-four 4096 blocks=$b"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --detector all --dist S --frames 1536 --steps 8 > $O/all1536_$b.json 2>/dev/null; line $O/all1536_$b.json "all-four S 1536 blocks=$b"
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --detector edges --dist T --frames 2048 --steps 4 --warmup 2 > $O/edgesT_$b.json 2>/dev/null; line $O/edgesT_$b.json "edges T 2048 blocks=$b"
done
