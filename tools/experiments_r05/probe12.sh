#!/bin/bash
# round 5, probe 12: which scramble of the stored V plane buys what probe 10 found (6 per-dword Weyl key, 7 one key per 128-B line,
# 8 one constant key, 9 Weyl key on the low byte of each dword; 4 = probe 10's best).  V-mode kernel time, 1024 x 1080p, S / T / K.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05m; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
cd /tmp; export TMPDIR=/tmp
for d in S T K; do for v in default vs8 kB kC kD kE kF kG default vs8; do
  L=$A/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  [ -f $L ] || continue
  rm -rf /tmp/etrace
  PSD_LIB_PATH=$L timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/etrace -o t --output-format csv -- python $R/tools/edge_ab.py 1024 $d $v > $O/ab_${v}_$d.log 2>&1
  python - <<PY
import csv
try:
    for r in csv.DictReader(open('/tmp/etrace/t_kernel_stats.csv')):
        n = r['Name']
        if 'score_frames_dma' in n: print('$v $d', n[:40], 'avg us', round(float(r['AverageNs'])/1e3,1), 'calls', r['Calls'])
except Exception as ex: print('$v $d no trace', ex)
PY
done; done 2>&1 | tee $O/vstore_keys.txt
