#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4m}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests/test_fullsize_models_gpu.py -q -s ) > $O/pytest_full.log 2>&1; grep -E "passed|failed|^E  |^\{" $O/pytest_full.log | cut -c1-700 | tail -12
cp gpurun_out/parity_whole_model_full_*.json $O/ 2>/dev/null
