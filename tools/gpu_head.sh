#!/bin/bash
# head kernels: tests + micro (one gpurun call).  usage: bash tools/gpu_head.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-head}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "mlp_head" ) > $O/pytest.log 2>&1; grep -E "passed|failed|^E  |^FAILED" $O/pytest.log | cut -c1-300 | tail -12
timeout 300 python tools/head_micro.py 2>$O/micro.err | tail -1 | tee $O/micro.json
tail -3 $O/micro.err
