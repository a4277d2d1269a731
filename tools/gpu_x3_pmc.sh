#!/bin/bash
# SQ counters + timings of the split-operand GEMM at the two dominant launch shapes (token GEMM, weight gradient)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02e}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for shape in "236672 384 128 0 0" "384 128 236672 1 1" "236672 256 128 0 1" "236672 128 256 0 0"; do
  for m in f32 bf16x3 bf16; do
    echo "== $shape $m" >> $O/timing.log
    GT_PRECISION=$m timeout 120 python tools/gemm_probe.py $shape 2>&1 | tail -1 >> $O/timing.log
  done
done
bash tools/gpu_pmc_sq.sh $TAG/sq_qkv 236672 384 128 0 0 > $O/sq_qkv.txt 2>&1
bash tools/gpu_pmc_sq.sh $TAG/sq_wgrad 384 128 236672 1 1 > $O/sq_wgrad.txt 2>&1
rm -rf $O/sq_qkv/sq $O/sq_wgrad/sq
cat $O/timing.log $O/sq_qkv.txt $O/sq_wgrad.txt
