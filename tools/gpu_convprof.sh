#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02conv}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/bench.log 2>&1
cd $R
MS=$(grep '^{"metric' $O/bench.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
python tools/prof_csv_summary.py $O/prof 80 --last-ms $MS --by-grid > $O/kernels_by_grid.txt 2>&1
rm -rf $O/prof
head -45 $O/kernels_by_grid.txt | cut -c1-180
