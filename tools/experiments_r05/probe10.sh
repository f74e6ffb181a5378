#!/bin/bash
# round 5, tenth probe: the open question of profiles/r04_ag_* -- why does the V-plane store of the V-mode front end cost 0.17 ms per
# 1024 shot-like frames and 0.04 ms on noise?  Store variants (PSD_VSTORE_MODE, built by `tools/next_vstore_ab.sh build`):
# 1 non-temporal, 2 bytes scrambled with position and frame (wrong edges), 3 sc0 sc1, 4 scrambled with the position only (wrong
# edges), 5 the plane stored as 255 - V (right edges).  V-mode kernel time from rocprofv3 --kernel-trace, 1024 x 1080p, S / K / U.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
A=$R/pyscenedetect_amd/csrc/build/abl
cd /tmp; export TMPDIR=/tmp
for d in S U K; do for v in default vs1 vs2 vs3 vs4 vs5 default; do
  [ $d = K ] && [ $v != default ] && [ $v != vs2 ] && [ $v != vs5 ] && continue
  L=$A/libpsd_$v.so; [ $v = default ] && L=$R/pyscenedetect_amd/libpsd_hip.so
  [ -f $L ] || continue
  rm -rf /tmp/etrace
  PSD_LIB_PATH=$L timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/etrace -o t --output-format csv -- python $R/tools/edge_ab.py 1024 $d $v > $O/ab_${v}_$d.log 2>&1
  grep "records crc" $O/ab_${v}_$d.log | cut -c1-200
  python - <<PY
import csv
try:
    for r in csv.DictReader(open('/tmp/etrace/t_kernel_stats.csv')):
        n = r['Name']
        if 'score_frames_dma' in n or 'sobel' in n or 'hysteresis' in n: print('$v $d', n[:40], 'avg us', round(float(r['AverageNs'])/1e3,1), 'calls', r['Calls'])
except Exception as ex: print('$v $d no trace', ex)
PY
done; done 2>&1 | tee $O/vstore_modes.txt
