#!/bin/bash
# energy per operation, plain vs packed fp32: tools/ubench/pk_power <mode> while rocm-smi samples power and clocks
O=gpurun_out/${1:-r02s}; mkdir -p $O
for m in fma pkfma add pkadd perm; do
  ./tools/ubench/pk_power $m 6 > $O/pk_$m.txt 2>&1 &
  BP=$!
  sleep 3.5
  for i in 1 2 3; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.4; done > $O/pk_smi_$m.txt
  wait $BP
  cat $O/pk_$m.txt $O/pk_smi_$m.txt
done | tee $O/pk_power.txt
