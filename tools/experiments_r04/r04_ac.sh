#!/bin/bash
# round 4, step ac: two barriers fewer per Sobel tile (clearing in front of phase 1's barrier; none before the store of a tile without candidates)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_ac; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
B=$R/pyscenedetect_amd/csrc/build/abl/libpsd_base.so
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_headline_geometry.py tests/test_gpu_parity.py tests/test_gpu_switches.py -m gpu -q -x --timeout=600 --timeout-method=thread -k "edge or hysteresis or dilation or serpentine or one_read or switch" > $O/pytest_edges.log 2>&1; echo "pytest rc=$?" >> $O/pytest_edges.log; tail -3 $O/pytest_edges.log
{ PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STU base; python tools/edge_ab.py 2048 STU new; PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STU base; python tools/edge_ab.py 2048 STU new; } 2>&1 | grep -v amdgpu.ids | tee $O/edge_ab.txt
