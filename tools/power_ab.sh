#!/bin/bash
# Socket power and clocks (rocm-smi, sampled while the kernel runs) during long runs of the default bench with different builds.
R=$PWD; O=gpurun_out/${1:-r02d}; mkdir -p $O; shift
for v in "$@"; do
  if [ $v = default ]; then L=$R/pyscenedetect_amd/libpsd_hip.so; else L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; fi
  PSD_LIB_PATH=$L python bench.py --no-cpu-baseline --no-secondary $BENCH_ARGS --steps ${STEPS:-2500} --warmup 5 > $O/bench_$v.json 2>/dev/null &
  BP=$!
  sleep 7
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|fclk|GPU use" | tr '\n' ' ' ; echo; sleep 0.7; done > $O/smi_$v.txt
  wait $BP
  echo "== $v: $(python -c "import json; d=json.load(open('$O/bench_$v.json')); print(d['value'], d['roofline']['avg_launch_ms'])")"
  cat $O/smi_$v.txt
done | tee $O/power.txt
