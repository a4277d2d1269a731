#!/bin/bash
# head f16x2 kernels in the step: module / whole-model parity tests + same-lease A/B.  usage: bash tools/gpu_head2.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-head2}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "mlp_head" ) > $O/pytest_k.log 2>&1; tail -1 $O/pytest_k.log
( timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_bench_kernels_gpu.py tests/test_train_gpu.py -q ) > $O/pytest_m.log 2>&1; grep -E "passed|failed|^FAILED|^E   " $O/pytest_m.log | cut -c1-250 | tail -12
BENCH_FAST="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy"
for V in 0 1 0 1; do
  GT_HEAD_F16=$V timeout 300 python bench.py $BENCH_FAST 2>/dev/null | tail -1 | python -c "import sys,json;r=json.loads(sys.stdin.read());print('GT_HEAD_F16=$V', r['value'], r['ms_per_step'])" | tee -a $O/bench_ab.txt
done
