#!/bin/bash
# quick validation: GPU suite + one short bench (+ optional steady-state kernel table).  usage: gpu_quick.sh <tag> [prof]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-quick}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/suite.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/suite.log | tail -12
if [ "$2" == "prof" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
      python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/bench.log 2>&1
  cd $R
  MS=$(grep '^{"metric' $O/bench.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
  python tools/prof_csv_summary.py $O/prof 50 --last-ms $MS > $O/kernels.txt 2>&1
  rm -rf $O/prof
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench.log; head -40 $O/kernels.txt | cut -c1-150
else
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*'
fi
