#!/bin/bash
# pJ per byte of the frame stream by access path: tools/ubench/stream_power <mode> while rocm-smi samples power and clocks
O=gpurun_out/${1:-r02e}; mkdir -p $O
for m in ${MODES:-x4 x3 dma}; do
  ./tools/ubench/stream_power $m 7 > $O/sp_$m.txt 2>&1 &
  BP=$!
  sleep 4
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done > $O/sp_smi_$m.txt
  wait $BP
  cat $O/sp_$m.txt $O/sp_smi_$m.txt
done | tee $O/stream_power.txt
