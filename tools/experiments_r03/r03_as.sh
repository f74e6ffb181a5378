#!/bin/bash
# feeder uploads tap rows only (psd_upload_rows) + cached NEAREST / AREA / hash tables: new GPU tests, the whole GPU suite, host-feed rates
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r03_as}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_feed_rows.py -m gpu -q -x --timeout=300 > $O/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -3 $O/pytest_new.log; grep -E "^E " $O/pytest_new.log | head -20
timeout 300 python tools/feed_bench.py > $O/host_feed_rates.json 2> $O/feed.err; echo "feed rc=$?"; cat $O/host_feed_rates.json; tail -3 $O/feed.err
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^(FAILED|ERROR)" $O/pytest.log | head -20; grep -E "^E " $O/pytest.log | head -20
