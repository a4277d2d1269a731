#!/bin/bash
# conv0 + resize micro runs over library variants.  usage: bash tools/gpu_crb_micro.sh <tag> [variant tags...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-crbm}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for V in "" "$@"; do
  GT_HIP_LIB=libgt_hip${V:+_$V}.so timeout 200 python tools/crb_micro.py 2>/dev/null | tail -1 | tee -a $O/micro.jsonl
done
