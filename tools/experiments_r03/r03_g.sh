#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_g}; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
PSD_LIB_PATH=$R/pyscenedetect_amd/csrc/build/abl/libpsd_$1.so timeout 900 python -m pytest tests -m gpu -q --timeout=600 > $O/pytest_$1.log 2>&1; echo "pytest($1) rc=$?"; grep -E "passed|failed" $O/pytest_$1.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest_$1.log | head
bash tools/experiments_r03/r03_ab.sh $T "$@"
