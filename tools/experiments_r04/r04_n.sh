#!/bin/bash
export PYTHONPATH=$PWD:$PWD/tools
run() { label=$1; shift; env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
run default X=1
run nch1 NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
run nch2 NCCL_MAX_NCHANNELS=2 NCCL_MIN_NCHANNELS=1
EXTRA="--exchange inline" run inline X=1
EXTRA="--exchange inline" run inline_nch1 NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
EXTRA="--exchange off" run off X=1
run default X=1
