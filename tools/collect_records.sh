#!/bin/bash
# Copy the records of one `tools/gpu_r6_final.sh <tag> ...` pass from gpurun_out/<tag>/ into profiles/ under the names
# profiles/README.md lists, and stamp profiles/SOURCE.json's hash from the pass.  usage: tools/collect_records.sh [tag]
set -e
T=${1:-r07}; S=gpurun_out/$T; P=profiles
cp $S/bench.json $P/r07_bench.json
cp $S/bench_table.json $P/r07_bench_table.json
cp $S/pmc_step.json $P/pmc_step.json
cp $S/pmc_summary.txt $P/r07_pmc_step_summary.txt
cp $S/kernel_stats_steady.txt $P/r07_rocprofv3_steady_B128_one_stream_by_grid.txt
cp $S/kernel_stats_whole_process.txt $P/r07_rocprofv3_kernel_stats_whole_process_no_f32_leg.txt
cp $S/sweep.json $P/r07_batch_sweep.json
for f in $S/parity_*.json; do cp $f $P/r07_$(basename $f); done
for W in ex1_burgers ex2_darcy211_fourier ex3_darcy_inv ex4_ns; do
  [ -f $S/bench_$W.json ] || continue
  cp $S/bench_$W.json $P/r07_bench_$W.json
  cp $S/kernel_stats_steady_$W.txt $P/r07_rocprofv3_steady_$W.txt
  cp $S/pmc_step_$W.json $P/pmc_step_$W.json
  cp $S/pmc_summary_$W.txt $P/r07_pmc_summary_$W.txt
done
[ -f $S/bench_weighted_l2.json ] && cp $S/bench_weighted_l2.json $P/r07_bench_weighted_l2.json
[ -f $S/fourier16_sq_counters.txt ] && cp $S/fourier16_sq_counters.txt $P/r07_fourier16_sq_counters.txt
echo "pass hash: $(cat $S/source_hash.txt)   tree hash: $(python -c 'import bench; print(bench.source_hash())')"
