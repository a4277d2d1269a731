#!/bin/bash
# sign-alternating accumulation: parity bisect (default arithmetic only), GEMM tests, quick bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4c}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time BISECT_QUICK=1 timeout 900 python tools/parity_bisect.py 9 $O/bisect_B9.json ) > $O/bisect.log 2>&1; echo "bisect rc=$?"; grep -E "^oracle|^bf16x3|^f32" $O/bisect.log
timeout 600 python tools/x3_bias_probe.py bf16x3 > $O/bias.jsonl 2> $O/bias.err; python - <<PY
import json
for l in open("$O/bias.jsonl"):
    r=json.loads(l); print(r['signs'], r['MNK'], r['split_k'], "mean %+.2e rms %.2e" % (r['bf16x3']['mean_signed'], r['bf16x3']['rms']))
PY
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or conv or scaler or x3" ) > $O/pytest_gemm.log 2>&1; tail -4 $O/pytest_gemm.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg --no-accuracy 2>$O/bench.err | tail -1 > $O/bench.json; python -c "import json;r=json.load(open('$O/bench.json'));print('bench',r['value'],r['ms_per_step'], r.get('roofline',{}).get('kernel_us'), r.get('roofline',{}).get('frac'))"
