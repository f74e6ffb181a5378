# SQ counters of the edge pipeline's kernels (tools/edge_time.py, shot-like content).  usage: bash tools/pmc_edges.sh OUTNAME
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OUT=$R/gpurun_out/$1/pmc_edges; mkdir -p $OUT
export ET_N=${ET_N:-256} ET_SMOOTH=1
( cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- python $R/tools/edge_time.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run grbm GRBM_GUI_ACTIVE GRBM_COUNT )
for k in sobel_nms value_plane dilate_xor hysteresis; do echo "== $k"; python tools/pmc_summary.py $OUT $k; done > $R/gpurun_out/$1/pmc_edges.txt
cat $R/gpurun_out/$1/pmc_edges.txt
