#!/bin/bash
# Round 6, C4 (ex3) off the library: bash tools/gpu_r6_c4.sh <tag> [part ...]    parts: ktests mtests bench prof
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r07c4}; shift
PARTS="${@:-ktests mtests bench prof}"
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
for part in $PARTS; do
  case $part in
    ktests)
      timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "silu or chain or conv3x3_resize or seg or resize" 2>&1 | tail -15 > $O/ktests.txt
      cat $O/ktests.txt;;
    mtests)
      timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_modules_gpu.py -x -q -k "no_library or darcy_inv or c4" 2>&1 | tail -15 > $O/mtests.txt
      cat $O/mtests.txt
      timeout 1200 python -m pytest tests/test_fullsize_models_gpu.py -x -q -k "ex3" 2>&1 | tail -15 > $O/fulltests.txt
      cat $O/fulltests.txt;;
    bench)
      timeout 600 python bench.py --workload ex3_darcy_inv --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg 2>$O/bench_c4.err | tail -1 > $O/bench_c4.json
      python -c "import json;r=json.load(open('$O/bench_c4.json'));print('C4',r['value'],r['ms_per_step'])" || tail -5 $O/bench_c4.err;;
    prof)
      cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o c4 --output-format csv -- python $R/bench.py --workload ex3_darcy_inv --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg > /dev/null 2>&1
      cd $R
      python tools/prof_csv_summary.py $O/prof_c4 60 --last-ms 350 --by-grid > $O/rocprofv3_c4.txt 2>/dev/null || ls -R $O/prof_c4 | head
      find $O/prof_c4 -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats.csv
      rm -rf $O/prof_c4
      head -45 $O/rocprofv3_c4.txt;;
  esac
done
