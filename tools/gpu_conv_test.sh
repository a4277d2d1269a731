#!/bin/bash
# implicit-GEMM convolution: its tests, then the step with and without it
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02cv}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3x3 or upsample or upscaler or x3 or gemm" > $O/tests.log 2>&1
tail -5 $O/tests.log
for v in 1; do
  GT_CONV_IMPLICIT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/bench_implicit$v.log 2>&1
  grep '^{"metric' $O/bench_implicit$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('implicit=$v', d['ms_per_step'], d['value'])"
done
