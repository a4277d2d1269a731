#!/bin/bash
# Round-4 probe call: MFMA rounding mode, signed error of the split-operand product, per-class bisect of the exact-math excess
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r4a; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 120 tools/_bin/mfma_round_probe > $O/mfma_round.json 2> $O/mfma_round.err; echo "mfma probe rc=$?"; head -c 1500 $O/mfma_round.json
timeout 600 python tools/x3_bias_probe.py > $O/bias.jsonl 2> $O/bias.err; echo "bias rc=$?"; tail -3 $O/bias.err
( time timeout 1500 python tools/parity_bisect.py 9 $O/bisect_B9.json ) > $O/bisect.log 2>&1; echo "bisect rc=$?"; tail -25 $O/bisect.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>$O/bench.err | tail -1 > $O/bench.json; python -c "import json;r=json.load(open('$O/bench.json'));print('bench',r['value'],r['ms_per_step'])"
