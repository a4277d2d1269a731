#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4n}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "f16x2" ) > $O/pytest_f16.log 2>&1; grep -E "passed|failed|^E  |rms" $O/pytest_f16.log | cut -c1-300 | tail -12
( time GT_PRECISION=f16x2 timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_fullsize_gpu.py -q -x ) > $O/pytest_mod_f16.log 2>&1; grep -E "passed|failed|^E  " $O/pytest_mod_f16.log | cut -c1-300 | tail -6
for P in bf16x3 f16x2; do timeout 300 python bench.py --precision $P --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>$O/bench_$P.err | tail -1 > $O/bench_$P.json; python -c "import json;r=json.load(open('$O/bench_$P.json'));print('bench $P',r['value'],r['ms_per_step'])"; done
( BISECT_QUICK=1 GT_PRECISION=f16x2 timeout 900 python tools/parity_bisect.py 9 $O/bisect_B9_f16x2.json ) > $O/bisect.log 2>&1; grep -E "^oracle|^bf16x3|^f32|^f16x2" $O/bisect.log | cut -c1-200
