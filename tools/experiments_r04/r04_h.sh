#!/bin/bash
out=gpurun_out/r04_h; mkdir -p $out
export PYTHONPATH=$PWD:$PWD/tools
python tools/flow_profile.py bbc > $out/flow_bbc.txt 2>&1; head -45 $out/flow_bbc.txt | cut -c1-150
python tools/flow_profile.py corpus > $out/flow_corpus.txt 2>&1; head -40 $out/flow_corpus.txt | cut -c1-150
