#!/bin/bash
# round 4, step af: equal chunks when a batch exceeds the edge workspace (4096 frames: 2 x 2048 instead of 3006 + 1090)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_af; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
B=$R/pyscenedetect_amd/csrc/build/abl/libpsd_base.so
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_switches.py tests/test_gpu_headline_geometry.py -m gpu -q -x --timeout=600 --timeout-method=thread -k "edge or chunks or switch" > $O/pytest_edges.log 2>&1; echo "pytest rc=$?" >> $O/pytest_edges.log; tail -3 $O/pytest_edges.log
{ PSD_LIB_PATH=$B python tools/edge_ab.py 4096 ST base; python tools/edge_ab.py 4096 ST new; PSD_LIB_PATH=$B python tools/edge_ab.py 4096 ST base; python tools/edge_ab.py 4096 ST new; } 2>&1 | grep -v amdgpu.ids | tee $O/edge_ab.txt
