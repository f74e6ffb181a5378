#!/bin/bash
# what do the Sobel kernel's stores cost?  (ablations: results are wrong on purpose)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_x; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R:$R/tools
for v in default sobel_nostore sobel_nz; do
  if [ $v = default ]; then L=$R/pyscenedetect_amd/libpsd_hip.so; else L=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$v.so; fi
  for d in S T; do
    PSD_LIB_PATH=$L timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$v$d -o t --output-format csv -- python $R/tools/edge_ab.py 1024 $d $v > $O/edge_ab_$v$d.txt 2>/dev/null
    echo "== $v $d"; tail -2 $O/edge_ab_$v$d.txt | cut -c1-200
    python $R/tools/kernel_stats_md.py $O/tr_$v$d/t_kernel_stats.csv "$v $d" 2>/dev/null | grep -E "sobel|hysteresis|dilate|score_frames|median" | cut -c1-60,100-170
    rm -rf $O/tr_$v$d
  done
done 2>&1 | tee $O/summary.txt
