#!/bin/bash
# round 5, fifth probe: video-sized fuzz cases, another small-shape seed, the per-frame breakdown again, a determinism soak
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
timeout 300 python tools/fuzz_gpu.py --big --seconds 170 --seed 3 > $O/fuzz_big_seed3.json 2> $O/fuzz_big.err; echo "fuzz big rc=$?"; cut -c1-1200 $O/fuzz_big_seed3.json
timeout 200 python tools/fuzz_gpu.py --seconds 90 --seed 11 > $O/fuzz_seed11.json 2> /dev/null; echo "fuzz rc=$?"; cut -c1-600 $O/fuzz_seed11.json
timeout 120 python tools/experiments_r05/per_frame_breakdown.py > $O/per_frame_breakdown.txt 2>&1; tail -1 $O/per_frame_breakdown.txt
SOAK_SECS=8 timeout 200 python tools/soak.py > $O/soak.txt 2>&1; tail -6 $O/soak.txt
