#!/bin/bash
# packed-B kernel: kernel tests, then the step with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02pk}; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > $O/tests.log 2>&1
tail -5 $O/tests.log
for v in 1 0; do
  GT_X3_PACKED=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/bench_packed$v.log 2>&1
  grep '^{"metric' $O/bench_packed$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('packed=$v', d['ms_per_step'], d['value'])"
done
