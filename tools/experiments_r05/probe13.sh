#!/bin/bash
# round 5, probe 13: the keyed V plane as built (EdgeGeom::vkey) -- edge / switch / flow GPU tests, then the whole edge term on
# S / T / K / U with PSD_EDGE_VKEY=1 (default) and 0 (plain), alternating, 2048 x 1080p like the bench's secondary lines.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_switches.py tests/test_gpu_flows.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest_edges.txt 2>&1; tail -3 $O/pytest_edges.txt
for rep in 1 2; do for k in 1 0; do
  PSD_EDGE_VKEY=$k timeout 200 python tools/edge_ab.py 2048 STKU vkey$k 2>&1 | grep "records crc"
done; done | tee $O/vkey_ab.txt
