#!/bin/bash
# HBM traffic of ONE GEMM launch shape (rocprofv3 --pmc, separate passes).  usage: gpu_pmc_gemm.sh tag M N K la lb
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$C -o pmc --output-format csv -- python $R/tools/gemm_probe.py "$@" > $O/$C.log 2>&1
done
cd $R
python tools/pmc_summary.py $O 10 | grep -i "calls\|gemm" | tee $O/pmc_summary.txt
grep "TF" $O/WRITE_SIZE.log
