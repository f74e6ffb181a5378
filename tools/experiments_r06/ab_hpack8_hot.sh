#!/bin/bash
# the same A/B after a minute of the power-capped headline kernel (does the slow state of the all-four instance come with heat?)
cd ${GRAFT_REPO_ROOT:-$PWD}
L=pyscenedetect_amd/csrc/build/abl/libpsd_hp16.so
t() { PSD_LIB_PATH=$2 python bench.py --no-cpu-baseline --no-secondary $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s' % '$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; }
echo "## cold"; for i in 1 2; do t pack8 $PWD/pyscenedetect_amd/libpsd_hip.so "--downscale auto --detector all"; t pack16 $PWD/$L "--downscale auto --detector all"; done
python bench.py --no-cpu-baseline --no-secondary --steps 8000 --warmup 3 2>/dev/null | tail -1 | cut -c1-120
echo "## after 40 s of the headline kernel"
for i in 1 2 3 4 5 6; do t pack8 $PWD/pyscenedetect_amd/libpsd_hip.so "--downscale auto --detector all"; t pack16 $PWD/$L "--downscale auto --detector all"; done
