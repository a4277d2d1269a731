#!/bin/bash
# rocprofv3 kernel trace of bench.py + steady-state summary.  usage: bash tools/gpu_prof.sh <tag> [bench args]
TAG=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $O/prof.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 70 --last-ms $MS > $O/kernel_stats_steady.txt 2>&1
python tools/prof_csv_summary.py $O/prof 40 > $O/kernel_stats.txt 2>&1
rm -f $O/prof/trace_kernel_trace.csv
cat $O/kernel_stats_steady.txt
