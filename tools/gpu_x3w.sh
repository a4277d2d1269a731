#!/bin/bash
# x3w prefetch depth / block order A/B (one gpurun call).  usage: bash tools/gpu_x3w.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x3w}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "weight_gradient or repeat_launch" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
( GT_X3W_PF=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "weight_gradient" ) > $O/pytest_pf1.log 2>&1; tail -1 $O/pytest_pf1.log
for PF in 1 2; do for MAP in 0 1; do
  GT_X3W_PF=$PF GT_X3W_MAP=$MAP timeout 300 python tools/x3w_micro.py 2>$O/micro_$PF$MAP.err | tail -1 | tee -a $O/micro.jsonl
done; done
BENCH_FAST="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy"
for V in "1 0" "2 0" "2 1" "2 2" "1 0" "2 2"; do set -- $V
  GT_X3W_PF=$1 GT_X3W_MAP=$2 timeout 300 python bench.py $BENCH_FAST 2>/dev/null | tail -1 | python -c "import sys,json;r=json.loads(sys.stdin.read());print('PF $1 MAP $2', r['value'], r['ms_per_step'])" | tee -a $O/bench_ab.txt
done
