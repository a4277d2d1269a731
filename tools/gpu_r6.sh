#!/bin/bash
# Round-6 measurement pass (gpu_r5.sh + "--strong-global-batch 0" on every run but the line of record) (one gpurun call).  usage: bash tools/gpu_r5.sh <tag> <step> [<step> ...]
#   wprof:<workload>[@B]  steady rocprofv3 kernel stats of an informational workload, by grid      wpmc:<workload>[@B]  its FETCH/WRITE/SQ passes
#   wbench:<workload>[@B] its bench line with the roofline leg      py:<script args>  python <script args> > $O/<script>.log      sh:<cmd>  bash -c
#   suite   the whole -m gpu suite (default arithmetic)          full    tests/test_fullsize_models_gpu.py (C3 / C4 / C5 at full size)
#   bench   the default bench.py line (+ per-kernel table)        prof    rocprofv3 kernel stats of the steady step, one stream, by grid
#   pmc     FETCH_SIZE / WRITE_SIZE / SQ passes -> pmc_step.json   sweep   batch sweep        work   the informational workloads
#   fullprof  rocprofv3 kernel stats of a whole bench.py process      smoke   __graft_entry__.smoke()
#   mode:<p> short bench in arithmetic <p>      quick:<-k expr> kernel/module tests matching      fetch   FETCH_SIZE pass only
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r07}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
BENCH_FAST="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0"
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $step"
  case $name in
    suite) ( time timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_fullsize_models_gpu.py ) > $O/suite.log 2>&1; grep -E "passed|failed|^FAILED" $O/suite.log | cut -c1-250 | tail -8
           cp gpurun_out/parity_*.json $O/ 2>/dev/null;;
    full) ( time timeout 2400 python -m pytest tests/test_fullsize_models_gpu.py -q ) > $O/full.log 2>&1; grep -E "passed|failed|^FAILED" $O/full.log | cut -c1-250 | tail -4
          cp gpurun_out/parity_whole_model_full_*.json $O/ 2>/dev/null;;
    bench) ( time timeout 1500 python bench.py --table $O/bench_table.json 2>$O/bench.err | tail -1 ) > $O/bench.json 2>$O/bench.time
           python -c "import json;r=json.loads(open('$O/bench.json').read().strip().splitlines()[0]);print('bench',r['value'],r['ms_per_step'],'roofline',r['roofline']['kernel'],r['roofline']['avg_launch_us'],r['roofline']['frac'],'cpu',r['cpu_baseline']['value'],'acc',r['accuracy']['hip']['val_rel_l2_mean'],r['accuracy']['welch_t_test'])";;
    quick) ( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_modules_gpu.py -q -k "$arg" ) > $O/quick.log 2>&1; grep -E "passed|failed|^FAILED|^E  " $O/quick.log | cut -c1-250 | tail -12;;
    fetch)
      cd /tmp
      timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc/FETCH_SIZE -o pmc --output-format csv -- \
          python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 > $O/pmc_FETCH_SIZE.log 2>&1
      cd $R
      python tools/pmc_summary.py $O/pmc 60 > $O/fetch_summary.txt 2>&1
      rm -rf $O/pmc
      grep -E "x3w|x3h|x3r" $O/fetch_summary.txt | cut -c1-170;;
    fullprof)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats -d $O/fprof -o trace --output-format csv -- \
          python $R/bench.py --no-cpu-baseline --no-accuracy --strong-global-batch 0 $arg > $O/fullprof.log 2>&1
      cd $R
      python tools/prof_csv_summary.py $O/fprof 80 > $O/kernel_stats_whole_process.txt 2>&1
      rm -rf $O/fprof
      head -8 $O/kernel_stats_whole_process.txt | cut -c1-150; grep -ci "igemm\|naive_conv\|miopen\|Cijk" $O/kernel_stats_whole_process.txt;;
    smoke) ( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2;;
    mode) ( timeout 400 python bench.py $BENCH_FAST --precision $arg 2>$O/bench_$arg.err | tail -1 ) > $O/bench_prec_$arg.json
          python -c "import json;r=json.load(open('$O/bench_prec_$arg.json'));print('precision $arg',r['value'],r['ms_per_step'])";;
    prof)
      cd /tmp
      GT_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
          python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 $arg > $O/prof.log 2>&1
      cd $R
      MS=$(grep '^{"metric' $O/prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
      python tools/prof_csv_summary.py $O/prof 90 --last-ms $MS --by-grid > $O/kernel_stats_steady${arg// /_}.txt 2>&1
      rm -rf $O/prof
      head -30 $O/kernel_stats_steady${arg// /_}.txt | cut -c1-150;;
    pmc)
      cd /tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc/$C -o pmc --output-format csv -- \
            python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 > $O/pmc_$C.log 2>&1
      done
      timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc/SQ -o pmc --output-format csv -- \
          python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 > $O/pmc_SQ.log 2>&1
      cd $R
      python tools/pmc_to_json.py $O/pmc $O/pmc_step.json
      python tools/pmc_summary.py $O/pmc 60 > $O/pmc_summary.txt 2>&1
      rm -rf $O/pmc
      head -24 $O/pmc_summary.txt | cut -c1-170;;
    sweep)
      echo "[" > $O/sweep.json
      for B in 4 16 64 128 256; do
        timeout 300 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --strong-global-batch 0 --no-f32-leg 2>/dev/null | tail -1 >> $O/sweep.json
        echo "," >> $O/sweep.json
      done
      timeout 300 python bench.py --batch 4 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --strong-global-batch 0 --no-f32-leg 2>/dev/null | tail -1 >> $O/sweep.json
      echo "]" >> $O/sweep.json
      python -c "
import json
for r in json.load(open('$O/sweep.json')): print('B', r['config']['per_gpu_batch'], 'graph' if r['config']['hip_graph'] else 'eager', r['value'], r['ms_per_step'])";;
    wprof|wpmc|wbench)
      W=${arg%%@*}; BA=""; [[ "$arg" == *@* ]] && BA="--batch ${arg#*@}"
      if [ $name = wbench ]; then
        timeout 500 python bench.py --workload $W $BA --steps 10 --warmup 3 --no-cpu-baseline --no-accuracy --strong-global-batch 0 --no-f32-leg --table $O/bench_table_$W.json 2>$O/wbench_$W.err | tail -1 > $O/bench_$W.json
        python -c "import json;r=json.load(open('$O/bench_$W.json'));rf=r.get('roofline') or {};print('$W',r['value'],r['ms_per_step'],r['config']['per_gpu_batch'],rf.get('kernel'),rf.get('avg_launch_us'),rf.get('bound'),rf.get('frac'))"
      elif [ $name = wprof ]; then
        cd /tmp
        GT_DUAL_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/wprof_$W -o trace --output-format csv -- \
            python $R/bench.py --workload $W $BA --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 > $O/wprof_$W.log 2>&1
        cd $R
        MS=$(grep '^{"metric' $O/wprof_$W.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
        python tools/prof_csv_summary.py $O/wprof_$W 70 --last-ms $MS --by-grid > $O/kernel_stats_steady_$W.txt 2>&1
        rm -rf $O/wprof_$W
        head -24 $O/kernel_stats_steady_$W.txt | cut -c1-160
      else
        cd /tmp
        for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
          timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/wpmc_$W/${C%% *} -o pmc --output-format csv -- \
              python $R/bench.py --workload $W $BA --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 > $O/wpmc_${W}_${C%% *}.log 2>&1
        done
        cd $R
        [ -d $O/wpmc_$W/SQ_VALU_MFMA_BUSY_CYCLES ] && mv $O/wpmc_$W/SQ_VALU_MFMA_BUSY_CYCLES $O/wpmc_$W/SQ
        python tools/pmc_to_json.py $O/wpmc_$W $O/pmc_step_$W.json
        python tools/pmc_summary.py $O/wpmc_$W 40 > $O/pmc_summary_$W.txt 2>&1
        rm -rf $O/wpmc_$W
        head -16 $O/pmc_summary_$W.txt | cut -c1-170
      fi;;
    py) ( timeout 900 python $arg ) > $O/py_$(basename ${arg%% *} .py).log 2>&1; tail -15 $O/py_$(basename ${arg%% *} .py).log | cut -c1-220;;
    sh) bash -c "$arg" 2>&1 | tail -20;;
    work)
      for W in ex3_darcy_inv ex2_darcy211_fourier ex4_ns ex1_burgers; do
        timeout 400 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --strong-global-batch 0 --no-f32-leg 2>$O/work_$W.err | tail -1 > $O/bench_$W.json
        python -c "import json;r=json.load(open('$O/bench_$W.json'));print('$W',r['value'],r['ms_per_step'],r['config']['per_gpu_batch'])"
      done
      timeout 400 python bench.py --loss weighted_l2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --strong-global-batch 0 --no-f32-leg 2>/dev/null | tail -1 > $O/bench_weighted_l2.json
      python -c "import json;r=json.load(open('$O/bench_weighted_l2.json'));print('weighted_l2',r['value'],r['ms_per_step'])";;
  esac
done
