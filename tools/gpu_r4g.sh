#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-r4g}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_nhwc or scaler" ) > $O/pytest_convw.log 2>&1; grep -E "passed|failed|^E  " $O/pytest_convw.log | cut -c1-300 | tail -6
bash tools/gpu_r3.sh ${1:-r4g} prof 2>&1 | grep -E "steady|convw|x3r_kernel<1, 1, 3, 3, 0, 2>" | cut -c1-140
