#!/bin/bash
# the hue wrap on pixel pairs (-DPSD_HPAIR=1, build/abl/libpsd_hpair.so by tools/build_variant.sh) against the default build: parity first, then interleaved timing
cd ${GRAFT_REPO_ROOT:-$PWD}; V=$PWD/pyscenedetect_amd/csrc/build/abl/libpsd_hpair.so
PSD_LIB_PATH=$V timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_geometry.py tests/test_gpu_fuzz.py tests/test_gpu_flows.py -m gpu -q -x 2>&1 | tail -3
for cfg in "" "--dist S" "--res 4k --frames 2048" "--frames 2048 --dist S"; do echo "## content $cfg"; for i in 1 2; do tools/ab_libs.sh "$cfg" hpair; done; done
