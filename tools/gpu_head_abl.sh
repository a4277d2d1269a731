#!/bin/bash
# head_bwd16_kernel timing ablations (build.py --ablate-head variants), one gpurun call.  usage: bash tools/gpu_head_abl.sh <tag> [variants...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-headabl}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
for V in "" "$@"; do
  LIB=libgt_hip${V:+_$V}.so
  GT_HIP_LIB=$LIB timeout 200 python tools/head_micro.py 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('$LIB', 'f16x2 fwd', r['f16x2']['fwd_us'], 'bwd', r['f16x2']['bwd_us'], '| f32 fwd', r['f32']['fwd_us'], 'bwd', r['f32']['bwd_us'], '| dw1 err', r['f16x2']['err']['dw1'])" | tee -a $O/abl.txt
done
