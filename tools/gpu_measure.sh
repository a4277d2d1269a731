#!/bin/bash
# Round-2 measurement pass (one gpurun call): GPU tests, headline bench (+table), rocprofv3 kernel stats of the same
# command, PMC traffic / MFMA-busy passes of an eager step, batch sweep + graph-vs-eager, informational workloads,
# rocFFT A/B probe.  usage: bash tools/gpu_measure.sh <tag> [what ...]   what in: tests bench prof pmc sweep work fft
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02m}; shift
WHAT=${*:-"tests bench prof pmc sweep work fft"}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then ( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; fi
if has bench; then ( time timeout 900 python bench.py --table $O/table.json ) > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-600; fi
if has prof; then   # one stream (GT_DUAL_STREAM=0): per-kernel averages of kernels that do not overlap
  cd /tmp
  GT_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
      python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/prof.log 2>&1
  cd $R
  MS=$(grep '^{"metric' $O/prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
  python tools/prof_csv_summary.py $O/prof 70 --last-ms $MS --by-grid > $O/kernel_stats_steady.txt 2>&1
  python tools/prof_csv_summary.py $O/prof 40 > $O/kernel_stats.txt 2>&1
  rm -rf $O/prof
  head -30 $O/kernel_stats_steady.txt | cut -c1-150
fi
if has pmc; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc/$C -o pmc --output-format csv -- \
        python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/pmc_$C.log 2>&1
  done
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc/SQ -o pmc --output-format csv -- \
      python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/pmc_SQ.log 2>&1
  cd $R
  python tools/pmc_to_json.py $O/pmc $O/pmc_step.json
  python tools/pmc_summary.py $O/pmc 40 > $O/pmc_summary.txt 2>&1
  rm -rf $O/pmc
  head -24 $O/pmc_summary.txt | cut -c1-160
fi
if has sweep; then
  echo "[" > $O/sweep.json
  for B in 4 16 64 128 256; do
    timeout 300 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy 2>/dev/null | tail -1 >> $O/sweep.json
    echo "," >> $O/sweep.json
  done
  timeout 300 python bench.py --batch 4 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy 2>/dev/null | tail -1 >> $O/sweep.json
  echo "]" >> $O/sweep.json
  python - <<PY
import json
for r in json.load(open("$O/sweep.json")):
    print(r["config"]["per_gpu_batch"], "graph" if r["config"]["hip_graph"] else "eager", r["value"], r["ms_per_step"], (r.get("f32_mfma_exact") or {}).get("value"))
PY
fi
if has work; then
  for w in ex2_darcy211_fourier ex3_darcy_inv ex4_ns ex1_burgers; do
    timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/bench_$w.json
    python -c "import json; r=json.load(open('$O/bench_$w.json')); print('$w', r['value'], r['ms_per_step'], (r.get('f32_mfma_exact') or {}).get('value'))"
  done
  timeout 400 python bench.py --loss weighted_l2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy 2>/dev/null | tail -1 > $O/bench_weighted_l2.json
  python -c "import json; r=json.load(open('$O/bench_weighted_l2.json')); print('weighted_l2', r['value'], r['ms_per_step'])"
fi
if has fft; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/fft -o trace --output-format csv -- python $R/tools/rocfft_probe.py > $O/rocfft_probe.log 2>&1
  cd $R
  python tools/prof_csv_summary.py $O/fft 30 > $O/rocfft_probe_kernels.txt 2>&1
  rm -rf $O/fft
  grep '^{' $O/rocfft_probe.log; head -16 $O/rocfft_probe_kernels.txt | cut -c1-150
fi
