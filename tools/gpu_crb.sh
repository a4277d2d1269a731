#!/bin/bash
# conv0 + resize with recorded decisions: tests + alternating step runs.  usage: bash tools/gpu_crb.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-crb}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "conv3x3_resize or scaler" ) > $O/pytest.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/pytest.log | cut -c1-300 | tail -8
BENCH_FAST="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy"
for V in 0 1 0 1; do
  GT_CRB_BITS=$V timeout 300 python bench.py $BENCH_FAST 2>/dev/null | tail -1 | python -c "import sys,json;r=json.loads(sys.stdin.read());print('GT_CRB_BITS=$V', r['value'], r['ms_per_step'])" | tee -a $O/bench_ab.txt
done
cd /tmp
GT_DUAL_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/prof.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 90 --last-ms $MS --by-grid > $O/kernel_stats_steady.txt 2>&1
rm -rf $O/prof
grep -E "conv_resize" $O/kernel_stats_steady.txt | cut -c1-150
