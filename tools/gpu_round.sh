#!/bin/bash
# One gpurun call: GPU parity tests, the bench line, and a rocprofv3 kernel trace of the same command.
# usage (from repo root on the GPU box): bash tools/gpu_round.sh [tag] [bench args...]
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py --table $O/table.json "$@" ) > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $O/prof.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 60 --last-ms $MS > $O/kernel_stats_steady.txt 2>&1
python tools/prof_csv_summary.py $O/prof 40 > $O/kernel_stats.txt 2>&1
rm -f $O/prof/trace_kernel_trace.csv
head -64 $O/kernel_stats_steady.txt
