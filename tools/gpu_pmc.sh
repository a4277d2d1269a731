#!/bin/bash
# HBM traffic counters (separate passes, as the microarch guide prescribes) for the kernels of one eager step.
# usage: bash tools/gpu_pmc.sh <tag> [bench args]
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d $O/$C -o pmc --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline "$@" > $O/$C.log 2>&1
  ls $O/$C | head
done
cd $R
python tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
head -50 $O/pmc_summary.txt
