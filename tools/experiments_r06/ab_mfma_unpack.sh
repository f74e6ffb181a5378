#!/bin/bash
# the staged HSV-only pass unpacking its pixels' bytes on the matrix core (-DPSD_MFMA_UNPACK=1, five / four waves per SIMD) against the default build
# and against the default at five waves per SIMD: parity first, then interleaved timing
cd ${GRAFT_REPO_ROOT:-$PWD}; D=$PWD/pyscenedetect_amd/csrc/build/abl
PSD_LIB_PATH=$D/libpsd_mfma5.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_geometry.py tests/test_gpu_fuzz.py tests/test_gpu_flows.py -m gpu -q -x 2>&1 | tail -3
for cfg in "" "--dist S" "--res 4k --frames 2048" "--frames 2048 --dist S" "--dist K --frames 2048"; do echo "## content $cfg"; for i in 1 2; do tools/ab_libs.sh "$cfg" mfma5 w5 mfma4; done; done
