#!/bin/bash
out=gpurun_out/r04_e; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_headline_geometry.py tests/test_gpu_switches.py -x -q -m gpu -k "edge or switch or golden or one_read" > $out/pytest.txt 2>&1
echo "pytest rc=$?" >> $out/pytest.txt; tail -12 $out/pytest.txt
bash tools/edge_trace.sh $out _sobel_list_vpart
