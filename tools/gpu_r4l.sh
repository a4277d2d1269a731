#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4l}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_nhwc or scaler or conv3x3" ) > $O/pytest_convw.log 2>&1; grep -E "passed|failed|^E  " $O/pytest_convw.log | cut -c1-300 | tail -6
for V in 1 0; do GT_CONV_WGRAD_PLANES=$V timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>$O/bench_$V.err | tail -1 > $O/bench_$V.json; python -c "import json;r=json.load(open('$O/bench_$V.json'));print('bench planes=$V',r['value'],r['ms_per_step'])"; done
bash tools/gpu_r3.sh ${1:-r4l} prof 2>&1 | grep -E "steady|convw|x3r_kernel<1, 1, 3, 3, 0, 2>" | cut -c1-140
