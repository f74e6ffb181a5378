# SQ counters of the fused all-detectors pass (bench.py --detector all) for the default library and optional ablation builds.
# usage (on the GPU box): bash tools/pmc_fused.sh OUTNAME [variant ...]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
NAME=$1; shift
for lib in default "$@"; do
  if [ $lib = default ]; then export PSD_LIB_PATH=$R/pyscenedetect_amd/libpsd_hip.so; else export PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$lib.so; fi
  OUT=$R/gpurun_out/$NAME/pmc_fused_$lib; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp
  run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc --output-format csv -- python $R/bench.py --detector all --no-cpu-baseline --no-secondary --frames 4096 --steps 2 --warmup 1 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM
  run grbm GRBM_GUI_ACTIVE GRBM_COUNT )
  python tools/pmc_summary.py $OUT score_frames > $R/gpurun_out/$NAME/pmc_fused_$lib.txt; cat $R/gpurun_out/$NAME/pmc_fused_$lib.txt
done
