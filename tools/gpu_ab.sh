#!/bin/bash
# A/B of one environment switch on the headline step, same box, back to back:  bash tools/gpu_ab.sh <tag> VAR [values...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ab}; mkdir -p $O
VAR=$2; shift 2
cd $R
for v in ${*:-1 0 1 0}; do
  env $VAR=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab_$VAR.log
done
