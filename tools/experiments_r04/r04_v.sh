#!/bin/bash
# round 4, step v: the increments of a quad go to four different histogram replicas (V mode; all-detectors pass as an experiment)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_v; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
B=$R/pyscenedetect_amd/csrc/build/abl/libpsd_base.so
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_headline_geometry.py tests/test_gpu_parity.py -m gpu -q -x --timeout=600 --timeout-method=thread > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
{ PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STUK base; python tools/edge_ab.py 2048 STUK new; PSD_LIB_PATH=$B python tools/edge_ab.py 2048 STUK base; python tools/edge_ab.py 2048 STUK new; } 2>&1 | grep -v amdgpu.ids | tee $O/edge_ab.txt
for d in S U K; do echo "all four fused, dist $d"; bash tools/ab_libs.sh "--detector all --dist $d" base rot; done 2>&1 | tee $O/fused_ab.txt
