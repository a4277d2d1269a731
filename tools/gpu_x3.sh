#!/bin/bash
# x3 bring-up: kernel tests of the split-operand GEMM, GPU suite in the default mode, bench + rocprofv3 per mode.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02c}; shift
MODES=${*:-"f32 bf16x3"}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm_x3" 2>&1 | tail -30 ) > $O/x3_tests.log 2>&1
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/suite_default.log 2>&1
for m in $MODES; do
  cd /tmp
  GT_PRECISION=$m timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$m -o trace --output-format csv -- \
      python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_$m.log 2>&1
  cd $R
  MS=$(grep '^{"metric' $O/bench_$m.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
  python tools/prof_csv_summary.py $O/prof_$m 45 --last-ms $MS > $O/kernels_$m.txt 2>&1
  rm -rf $O/prof_$m
done
tail -8 $O/x3_tests.log; echo "== suite"; tail -25 $O/suite_default.log
for m in $MODES; do echo "== bench $m"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_$m.log || tail -5 $O/bench_$m.log; head -40 $O/kernels_$m.txt; done
