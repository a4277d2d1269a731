#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4d}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_nhwc or scaler or conv3x3" ) > $O/pytest_convw.log 2>&1; grep -E "passed|failed|^E  " $O/pytest_convw.log | cut -c1-300 | tail -12
( time timeout 1500 python -m pytest tests/test_bench_kernels_gpu.py -q -x -k "whole_model" -s ) > $O/pytest_model.log 2>&1; grep -E "passed|failed|^E  |worst" $O/pytest_model.log | cut -c1-400 | tail -12
for V in hip miopen; do GT_SCALER_WGRAD=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>$O/bench_$V.err | tail -1 > $O/bench_$V.json; python -c "import json;r=json.load(open('$O/bench_$V.json'));print('bench $V',r['value'],r['ms_per_step'])"; done
