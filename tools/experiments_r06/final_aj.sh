# fourth session of round 6: the differential fuzz on the round's final tree (three seeds + video-sized frames), GPU suite, smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_aj; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for s in 71 72 73; do timeout 260 python tools/fuzz_gpu.py --seconds 200 --seed $s 2>/dev/null | tail -1 > $O/fuzz_gpu_seed$s.json; python -c "import json;d=json.load(open('$O/fuzz_gpu_seed$s.json'));print($s,{k:v for k,v in d.items() if k!='mismatches'} , len(d.get('mismatches',[])))"; done
timeout 200 python tools/fuzz_gpu.py --seconds 140 --seed 74 --big 2>/dev/null | tail -1 > $O/fuzz_gpu_big_seed74.json; python -c "import json;d=json.load(open('$O/fuzz_gpu_big_seed74.json'));print(74,{k:v for k,v in d.items() if k!='mismatches'} , len(d.get('mismatches',[])))"
