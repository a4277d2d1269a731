#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02f}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > $O/suite.log 2>&1
( timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -3 ) > $O/bench_default.log 2>&1
( GT_X3_RING_DEPTH=3 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg 2>&1 | tail -3 ) > $O/bench_depth3.log 2>&1
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --optimizer torch 2>&1 | tail -3 ) > $O/bench_torchopt.log 2>&1
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --loss weighted_l2 2>&1 | tail -3 ) > $O/bench_wl2.log 2>&1
for w in ex4_ns ex1_burgers ex3_darcy_inv; do
  ( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --workload $w 2>&1 | tail -3 ) > $O/bench_$w.log 2>&1
done
tail -12 $O/suite.log
for f in default depth3 torchopt wl2 ex4_ns ex1_burgers ex3_darcy_inv; do echo "== $f"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"f32_mfma_exact": {[^}]*}\|"final_loss": [0-9.e-]*' $O/bench_$f.log || tail -5 $O/bench_$f.log; done
grep -o '"roofline": {.*' $O/bench_default.log | cut -c1-1500
