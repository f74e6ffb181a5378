#!/bin/bash
# round 4, step aa: waves per workgroup of the per-frame hysteresis kernel, forced (PSD_EDGE_HF), 2048 and 1024 frames per submission
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04_aa; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
for n in 2048 1024; do for hf in 0 4 8 16 0; do PSD_EDGE_HF=$hf python tools/edge_ab.py $n TU hf$hf; done; done 2>&1 | grep -v amdgpu.ids | tee $O/edge_hf.txt
