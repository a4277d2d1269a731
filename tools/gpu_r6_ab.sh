#!/bin/bash
# Same-box A/B of environment switches on the headline line: bash tools/gpu_r6_ab.sh <tag> "<env a>" "<env b>" ...
# (each argument is a space-separated list of VAR=value; "-" = defaults).  Two rounds, interleaved, to see the box drift.
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-ab}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for round in 1 2; do
  for cfg in "$@"; do
    envs=""; [ "$cfg" != "-" ] && envs="$cfg"
    line=$(env $envs timeout 400 python bench.py --steps ${AB_STEPS:-20} --warmup 5 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg --strong-global-batch 0 ${AB_ARGS:-} 2>/dev/null | tail -1)
    echo "$line" > "$O/ab_${round}_$(echo "$cfg" | tr ' =' '__').json"
    python -c "import json,sys;r=json.loads(sys.argv[1]);print('round $round [$cfg]',r['value'],r['ms_per_step'])" "$line"
  done
done
