#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x3w2}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "weight_gradient" ) > $O/pytest.log 2>&1; tail -1 $O/pytest.log
timeout 300 python tools/x3w_micro.py 2>/dev/null | tail -1 | tee -a $O/micro.jsonl
GT_X3W_PF=1 timeout 300 python tools/x3w_micro.py 2>/dev/null | tail -1 | tee -a $O/micro.jsonl
