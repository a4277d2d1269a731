#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02h}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > $O/suite.log 2>&1
( GT_X3_VARIANT=ring timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "x3 or headnorm" 2>&1 | tail -5 ) > $O/suite_ring.log 2>&1
( GT_X3_VARIANT=ring timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>&1 | tail -2 ) > $O/bench_ring.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/bench_pipe.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/bench_pipe.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 30 --last-ms $MS > $O/kernels.txt 2>&1
rm -rf $O/prof
grep -E "passed|failed|FAILED|Error" $O/suite.log | tail -12; tail -2 $O/suite_ring.log
for f in ring pipe; do echo "== $f"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_$f.log || tail -5 $O/bench_$f.log; done
head -16 $O/kernels.txt | cut -c1-160
