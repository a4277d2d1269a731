#!/bin/bash
# Round-3 measurement driver (one gpurun call).  usage: bash tools/gpu_r3.sh <tag> <step> [<step> ...]
#   newtests   tests/test_bench_kernels_gpu.py (oracle parity on the bench's kernel instances)
#   suite      the whole -m gpu suite
#   micro      tools/x3_micro.py (token GEMMs in isolation)           micro:<lib>  with GT_HIP_LIB=<lib>
#   bench      short headline bench (graph)                            bench:<VAR=val,...>  with extra environment
#   prof       rocprofv3 kernel stats of the steady step, one stream, by grid
#   pmc        FETCH_SIZE / WRITE_SIZE / SQ passes -> pmc_step.json
#   full       the default bench.py line (+ table)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
BENCH_FAST="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy"
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $step"
  case $name in
    newtests) ( time timeout 1500 python -m pytest tests/test_bench_kernels_gpu.py -q -s ) > $O/newtests.log 2>&1; tail -5 $O/newtests.log;;
    suite) ( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/suite.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/suite.log | tail -15;;
    pytest) L=$O/pytest_$(echo $arg | tr '/:. ' '____' | cut -c1-60).log; ( time timeout 1200 python -m pytest $arg -q -x ) > $L 2>&1; grep -E "passed|failed|^E  " $L | cut -c1-300 | tail -n 8;;
    micro) ( env $(echo $arg | tr ',' ' ') timeout 300 python tools/x3_micro.py 2>>$O/micro.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); r['env']='$arg'; print(json.dumps(r))" ) >> $O/micro.jsonl
           python -c "import sys,json; r=json.loads(open('$O/micro.jsonl').read().strip().splitlines()[-1]); print('$arg', ' | '.join('%s %.0f' % (k[:18], v['us']) for k, v in r.items() if isinstance(v, dict)))";;
    benchp) ( timeout 400 python bench.py $BENCH_FAST --precision $arg 2>$O/bench.err | tail -1 ) > $O/bench_prec_$arg.json
           python -c "import json,sys; r=json.load(open(sys.argv[1])); print('precision $arg', r['value'], r['ms_per_step'], r['config']['final_loss'])" $O/bench_prec_$arg.json;;
    bench) ( env $(echo $arg | tr ',' ' ') timeout 400 python bench.py $BENCH_FAST 2>$O/bench.err | tail -1 ) > $O/bench_$(echo "$arg" | tr '=,/ ' '____').json
           python -c "import json,sys; r=json.load(open(sys.argv[1])); print('$arg', r['value'], r['ms_per_step'])" $O/bench_$(echo "$arg" | tr '=,/ ' '____').json;;
    prof)
      cd /tmp
      env $(echo $arg | tr ',' ' ') GT_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- \
          python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/prof.log 2>&1
      cd $R
      MS=$(grep '^{"metric' $O/prof.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step']*10)")
      python tools/prof_csv_summary.py $O/prof 80 --last-ms $MS --by-grid > $O/kernel_stats_steady.txt 2>&1
      rm -rf $O/prof
      head -45 $O/kernel_stats_steady.txt | cut -c1-160;;
    pmc)
      cd /tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc/$C -o pmc --output-format csv -- \
            python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/pmc_$C.log 2>&1
      done
      timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc/SQ -o pmc --output-format csv -- \
          python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy > $O/pmc_SQ.log 2>&1
      cd $R
      python tools/pmc_to_json.py $O/pmc $O/pmc_step.json
      python tools/pmc_summary.py $O/pmc 50 > $O/pmc_summary.txt 2>&1
      rm -rf $O/pmc
      head -30 $O/pmc_summary.txt | cut -c1-170;;
    sweep)
      echo "[" > $O/sweep.json
      for B in 4 16 64 128 256; do
        timeout 300 python bench.py --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg 2>/dev/null | tail -1 >> $O/sweep.json
        echo "," >> $O/sweep.json
      done
      timeout 300 python bench.py --batch 4 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg 2>/dev/null | tail -1 >> $O/sweep.json
      echo "]" >> $O/sweep.json
      python -c "
import json
for r in json.load(open('$O/sweep.json')): print(r['config']['per_gpu_batch'], 'graph' if r['config']['hip_graph'] else 'eager', r['value'], r['ms_per_step'])";;
    work)
      for w in ex2_darcy211_fourier ex3_darcy_inv ex4_ns ex1_burgers; do
        timeout 400 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-f32-leg 2>/dev/null | tail -1 > $O/bench_$w.json
        python -c "import json; r=json.load(open('$O/bench_$w.json')); print('$w', r['value'], r['ms_per_step'])"
      done
      timeout 400 python bench.py --loss weighted_l2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg 2>/dev/null | tail -1 > $O/bench_weighted_l2.json
      python -c "import json; r=json.load(open('$O/bench_weighted_l2.json')); print('weighted_l2', r['value'], r['ms_per_step'])";;
    fullprof)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace --stats -d $O/fprof -o trace --output-format csv -- \
          python $R/bench.py --no-cpu-baseline --no-accuracy > $O/fullprof.log 2>&1
      cd $R
      python tools/prof_csv_summary.py $O/fprof 60 > $O/kernel_stats_whole_process.txt 2>&1
      rm -rf $O/fprof
      head -12 $O/kernel_stats_whole_process.txt | cut -c1-150;;
    full) ( time timeout 900 python bench.py --table $O/table.json ) > $O/bench_full.log 2> $O/bench_full.err; tail -1 $O/bench_full.log | cut -c1-700;;
    probe) ( env $(echo $arg | tr ',' ' ') timeout 300 python tools/parity_probe.py 18 2>>$O/probe.err | tail -1 ) >> $O/probe.jsonl; tail -1 $O/probe.jsonl | cut -c1-900;;
    probe64) ( env $(echo $arg | tr ',' ' ') timeout 300 python tools/parity_probe.py 18 f64 2>>$O/probe.err | tail -1 ) >> $O/probe.jsonl; tail -1 $O/probe.jsonl | cut -c1-900;;
    *) echo "unknown step $step";;
  esac
done
