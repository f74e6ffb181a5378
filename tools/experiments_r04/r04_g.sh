#!/bin/bash
out=gpurun_out/r04_g; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_headline_geometry.py -x -q -m gpu -k "edge or golden or one_read" > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt; tail -6 $out/pytest.txt
bash tools/edge_trace.sh $out _hybrid_wake
