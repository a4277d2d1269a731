#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02g}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/suite.log 2>&1
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -3 ) > $O/bench_default.log 2>&1
( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --workload ex1_burgers 2>&1 | tail -3 ) > $O/bench_ex1_burgers.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg > $O/bench_prof.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/bench_prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 60 --last-ms $MS > $O/kernels.txt 2>&1
rm -rf $O/prof
grep -E "passed|failed|FAILED|Error" $O/suite.log | tail -12
for f in default ex1_burgers; do echo "== $f"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"f32_mfma_exact": {[^}]*}' $O/bench_$f.log || tail -5 $O/bench_$f.log; done
head -45 $O/kernels.txt | cut -c1-160
