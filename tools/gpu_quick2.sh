#!/bin/bash
# quick check: f16x2 kernel tests + short bench + profile top lines.  usage: gpu_quick2.sh <tag> [grep pattern for the profile]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-q}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "f16x2 or repeat_launch or wgrad or scaler or conv3x3" ) > $O/pytest.log 2>&1; grep -E "passed|failed|^E  " $O/pytest.log | cut -c1-300 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>$O/bench.err | tail -1 > $O/bench.json; python -c "import json;r=json.load(open('$O/bench.json'));print('bench',r['value'],r['ms_per_step'])"
bash tools/gpu_r3.sh ${1:-q} prof 2>&1 | grep -E "steady|${2:-pack}" | cut -c1-150
