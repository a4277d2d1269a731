#!/bin/bash
# Round-6 final measurement pass (one gpurun call): bash tools/gpu_r6_final.sh <tag> [part ...]     parts: tests head work fourier
#   tests    the whole -m gpu suite + the full-size model tests; parity records -> profiles/r07_parity_*.json (on the box) so that
#            the bench line quotes THIS build's records
#   head     headline: counter passes -> profiles/pmc_step.json (on the box), then the default bench.py line + per-shape table
#            (its roofline / legs read that counter file), steady by-grid rocprofv3 stats, whole-process stats, batch sweep
#   work     C3 / C5 / C4 / C1: counter passes -> profiles/pmc_step_<workload>.json, bench line with the workload's own dominant-
#            kernel roofline, steady by-grid stats
#   fourier  SQ counters of the gt_fourier16 kernels at C3's layer shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r07}; shift
PARTS="${@:-tests head work fourier}"
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
# the records of this pass belong to THESE sources: stamp the hash before any line quotes it (bench.py compares it with the sources it runs)
python - <<'PY' > $O/source_hash.txt
import json, bench
h = bench.source_hash()
try:
    rec = json.load(open("profiles/SOURCE.json"))
except Exception:
    rec = {}
rec["source_sha16"] = h
json.dump(rec, open("profiles/SOURCE.json", "w"), indent=1)
print(h)
PY
for part in $PARTS; do
  case $part in
    tests)
      bash tools/gpu_r6.sh $TAG suite full
      for f in gpurun_out/parity_*.json; do [ -f "$f" ] && cp $f profiles/r07_$(basename $f); done;;
    head)
      bash tools/gpu_r6.sh $TAG pmc
      cp $O/pmc_step.json profiles/pmc_step.json
      bash tools/gpu_r6.sh $TAG bench prof "fullprof:--no-f32-leg" sweep;;
    work)
      for W in ex2_darcy211_fourier ex4_ns ex3_darcy_inv ex1_burgers; do
        bash tools/gpu_r6.sh $TAG wpmc:$W
        cp $O/pmc_step_$W.json profiles/pmc_step_$W.json
        bash tools/gpu_r6.sh $TAG wbench:$W wprof:$W
      done
      timeout 400 python bench.py --loss weighted_l2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accuracy --no-f32-leg --strong-global-batch 0 2>/dev/null | tail -1 > $O/bench_weighted_l2.json
      python -c "import json;r=json.load(open('$O/bench_weighted_l2.json'));print('weighted_l2',r['value'],r['ms_per_step'])";;
    fourier)
      cd /tmp
      for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"; do
        timeout 200 rocprofv3 --pmc $C --kernel-trace -d $O/f16pmc/p_${C%% *} -o pmc --output-format csv -- python $R/tools/fourier16_drive.py > /dev/null 2>&1
      done
      cd $R
      python tools/pmc_kernel_table.py $O/f16pmc fourier16_kernel > $O/fourier16_sq_counters.txt
      rm -rf $O/f16pmc
      cat $O/fourier16_sq_counters.txt | head -40;;
  esac
done
