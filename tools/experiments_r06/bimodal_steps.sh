cd $GRAFT_REPO_ROOT
run() { python bench.py --downscale auto --detector all --no-secondary --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline'].get('box') or {}; print('$*', d['steps'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], b.get('shader_clock_mhz_median'), b.get('socket_power_w_median'))"; }
for i in 1 2 3; do run; run --steps 10; run --frames 4096 --steps 10 --warmup 3; run --steps 40; run --steps 5; done
