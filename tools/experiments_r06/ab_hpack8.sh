#!/bin/bash
# A/B of the partial-histogram layout of the fused downscale kernel (8-bit counts + escapes against 16-bit counts), interleaved, one box.
cd ${GRAFT_REPO_ROOT:-$PWD}
L=pyscenedetect_amd/csrc/build/abl/libpsd_hp16.so
t() { PSD_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-secondary $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % '$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
for cfg in "--downscale auto --detector all" "--downscale auto --detector all --dist S" "--workload corpus --steps 6 --warmup 3" "--downscale auto --detector edges --dist T --frames 4096" "--downscale auto --detector hist"; do
  echo "## $cfg"
  for i in 1 2 3; do t pack8 $PWD/pyscenedetect_amd/libpsd_hip.so "$cfg"; t pack16 $PWD/$L "$cfg"; done
done
