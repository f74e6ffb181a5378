#!/bin/bash
# PMC passes over a short bench run (counters only with --kernel-trace; one pass per rocprofv3 run).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
BENCH_ARGS="${@:---frames 4096 --steps 2 --warmup 1}"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
[ -n "$PMC_SKIP_SQ2" ] || run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
ls -R $OUT | head -40
