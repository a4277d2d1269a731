#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4o}; mkdir -p $O; cd $R; export TMPDIR=/tmp
for P in bf16x3 f16x2; do timeout 300 python bench.py --precision $P --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>$O/bench_$P.err | tail -1 > $O/bench_$P.json; python -c "import json;r=json.load(open('$O/bench_$P.json'));print('bench $P',r['value'],r['ms_per_step'], r['config'].get('final_loss'))"; done
( time timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_fullsize_models_gpu.py ) > $O/suite.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/suite.log | cut -c1-300 | tail -8
