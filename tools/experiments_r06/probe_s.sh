#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_s; mkdir -p $O; cd $R; export PYTHONPATH=$R:$R/tools
for i in 1 2 3 4; do python tools/experiments_r06/luma_bimodal_probe.py 4096; done > $O/bimodal.txt 2>&1; cat $O/bimodal.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_e -o t --output-format csv -- python $R/bench.py --downscale auto --detector edges --dist T --frames 4096 --no-secondary --no-cpu-baseline --steps 6 --warmup 2 > $O/bench_ds_edges_T.json 2>/dev/null
python $R/tools/kernel_stats_md.py $O/trace_e/t_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --downscale auto --detector edges --dist T --frames 4096 --steps 6 --warmup 2" > $O/kernel_trace_downscale_edges_T.md; cut -c1-220 $O/kernel_trace_downscale_edges_T.md | head -30; rm -rf $O/trace_e
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_a -o t --output-format csv -- python $R/bench.py --downscale auto --detector all --frames 4096 --no-secondary --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_ds_all.json 2>/dev/null
python $R/tools/kernel_stats_md.py $O/trace_a/t_kernel_stats.csv "all four behind the downscale" > $O/kernel_trace_downscale_all.md; cut -c1-220 $O/kernel_trace_downscale_all.md | head -14; rm -rf $O/trace_a
