#!/bin/bash
# Round 6: bash tools/gpu_r6_suite.sh <tag> [part ...]    parts: suite full wbench:<workload>[:B] headline
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r07}; shift
PARTS="${@:-suite}"
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
for part in $PARTS; do
  case $part in
    suite)
      timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_fullsize_models_gpu.py 2>&1 | tail -25 > $O/suite.txt
      cat $O/suite.txt;;
    full)
      timeout 1800 python -m pytest tests/test_fullsize_models_gpu.py -q -m gpu 2>&1 | tail -25 > $O/full.txt
      cat $O/full.txt;;
    wbench:*)
      IFS=: read _ W B <<< "$part"
      timeout 600 python bench.py --workload $W ${B:+--batch $B} --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg 2>$O/bench_$W.err | tail -1 > $O/bench_$W.json
      python -c "import json;r=json.load(open('$O/bench_$W.json'));print('$W',r['value'],r['ms_per_step'])" || tail -5 $O/bench_$W.err;;
    headline)
      timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg 2>$O/bench_head.err | tail -1 > $O/bench_head.json
      python -c "import json;r=json.load(open('$O/bench_head.json'));print('headline',r['value'],r['ms_per_step'])" || tail -5 $O/bench_head.err;;
    wprof:*)
      IFS=: read _ W B <<< "$part"
      cd /tmp
      GT_DUAL_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$W -o p --output-format csv -- python $R/bench.py --workload $W ${B:+--batch $B} --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg --strong-global-batch 0 > /dev/null 2>&1
      cd $R
      python tools/prof_csv_summary.py $O/prof_$W 70 --last-ms 300 --by-grid > $O/rocprofv3_$W.txt 2>/dev/null
      rm -rf $O/prof_$W
      head -50 $O/rocprofv3_$W.txt;;
  esac
done
