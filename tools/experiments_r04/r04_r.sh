#!/bin/bash
# round 4, step r: Sobel candidates classified without a branch per direction (no extra LDS); counters of the hysteresis kernel on noise
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_r; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
B=$R/pyscenedetect_amd/csrc/build/abl/libpsd_base.so
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_headline_geometry.py tests/test_gpu_parity.py tests/test_gpu_flows.py -m gpu -q -x --timeout=600 --timeout-method=thread -k "edge or hysteresis or dilation or serpentine or corpus or one_read" > $O/pytest_edges.log 2>&1; echo "pytest rc=$?" >> $O/pytest_edges.log; tail -3 $O/pytest_edges.log
{ PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STU base; python tools/edge_ab.py 2048 STU new; PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STU base; python tools/edge_ab.py 2048 STU new; } 2>&1 | grep -v amdgpu.ids | tee $O/edge_ab.txt
cd /tmp; export TMPDIR=/tmp
P=$O/pmc_U; mkdir -p $P
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $P/$name -o pmc --output-format csv -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 2 --warmup 1 --detector edges --dist U --frames 1024 > $P/$name.log 2>&1; echo "$name rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
run tcc1 FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcc2 TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
run tcc3 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
python $R/tools/pmc_by_kernel.py $P hysteresis sobel_nms > $O/pmc_U_hysteresis_sobel.txt 2>&1; cat $O/pmc_U_hysteresis_sobel.txt | tr -s ' ' | sed 's/  /\n    /g' | cut -c1-200
rm -rf $P/*/
