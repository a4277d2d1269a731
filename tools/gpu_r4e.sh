#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4e}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "wgrad_nhwc or scaler or conv3x3" ) > $O/pytest_convw.log 2>&1; grep -E "passed|failed|^E  " $O/pytest_convw.log | cut -c1-300 | tail -6
( time timeout 1500 python -m pytest tests/test_bench_kernels_gpu.py -q -k "whole_model" -s ) > $O/pytest_model.log 2>&1; grep -E "passed|failed|^E  |worst" $O/pytest_model.log | cut -c1-600 | tail -12
cp gpurun_out/parity_whole_model_*.json $O/ 2>/dev/null
