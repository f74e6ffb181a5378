#!/bin/bash
# V-mode front end experiments: kernel times (rocprofv3 kernel trace) of the edge + HSV submission on 1024 x 1080p S frames
out=gpurun_out/r04_f; mkdir -p $out
R=$PWD; export PYTHONPATH=$R:$R/tools
cd /tmp; export TMPDIR=/tmp
run() {  # name, lib, env...
  name=$1; lib=$2; shift; shift
  rm -rf /tmp/etr
  env PSD_LIB_PATH=$lib "$@" timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/etr -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 --detector edges --dist ${DIST:-S} --frames 1024 > /tmp/b.json 2>/dev/null
  python $R/tools/kernel_stats_md.py /tmp/etr/t_kernel_stats.csv "$name" > $R/$out/trace_$name.md 2>&1
  echo "== $name: $(python -c "import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['roofline']['avg_launch_ms'])")"
  grep -E "score_frames|v_hist|vpart|sobel|hysteresis_frame" $R/$out/trace_$name.md | cut -c1-60,95-150
}
D=$R/pyscenedetect_amd/libpsd_hip.so; A=$R/pyscenedetect_amd/csrc/build/abl
run default $D X=1
run fused16 $D PSD_EDGE_VHIST_FUSED=1
run vm_vhist $D PSD_EDGE_VHIST_FUSED=0
run vhf1 $A/libpsd_vhf1.so X=1
run vhnoadd $A/libpsd_vhnoadd.so X=1
run vhnostore $A/libpsd_vhnostore.so X=1
run vhnone $A/libpsd_vhnone.so X=1
DIST=T run default_T $D X=1
DIST=U run default_U $D X=1
for mb in 256 512 1024; do
  run default_ws$mb $D PSD_EDGE_WS_MB=$mb
  run fused16_ws$mb $D PSD_EDGE_VHIST_FUSED=1 PSD_EDGE_WS_MB=$mb
done
