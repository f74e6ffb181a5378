#!/bin/bash
# PMC passes over the HashDetector thumbnail kernel (1024 x 1080p, S = 16).  usage: tools/pmc_hash.sh <outdir>
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- python $R/tools/hash_time.py 1024 1080p 16 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
