mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "exhaustive or records_match or padded or constant or randomised or full_size or 4k_batch or shot_like" 2>&1 | tail -5 > gpurun_out/r02b/pytest_fp.log; cat gpurun_out/r02b/pytest_fp.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 ', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
PSD_LIB_PATH=$PWD/pyscenedetect_amd/csrc/build/abl/libpsd_int.so python bench.py --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('int  ', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
