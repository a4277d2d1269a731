#!/bin/bash
# A/B of library variants (GT_HIP_LIB) on the GEMM micro-benchmark and the headline step:  bash tools/gpu_lib_ab.sh <tag> lib...
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-libab}; mkdir -p $O; shift
cd $R
for rep in 1 2; do for lib in "$@"; do
  GT_HIP_LIB=$lib timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['value'])" | tee -a $O/lib_ab.log
done; done
for lib in "$@"; do GT_HIP_LIB=$lib timeout 200 python tools/x3_micro.py 2>/dev/null | tail -1 | tee -a $O/micro.jsonl | cut -c1-900; done
