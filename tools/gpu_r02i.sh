#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02i}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/suite.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/bench.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/bench.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 30 --last-ms $MS > $O/kernels.txt 2>&1
rm -rf $O/prof
cd /tmp
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc/WRITE_SIZE -o pmc --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/pmc_w.log 2>&1
cd $R
python tools/pmc_summary.py $O/pmc 24 > $O/pmc_summary.txt 2>&1; rm -rf $O/pmc
grep -E "passed|failed|FAILED|Error" $O/suite.log | tail -12
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench.log
head -14 $O/kernels.txt | cut -c1-150; grep x3 $O/pmc_summary.txt | cut -c1-150
