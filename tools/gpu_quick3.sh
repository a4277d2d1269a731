#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-q3}; mkdir -p $O; cd $R; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_nhwc or scaler" ) > $O/pytest.log 2>&1; grep -E "passed|failed|^E  " $O/pytest.log | cut -c1-300 | tail -3
for C in default 2 3 4; do
  export GT_CW_CIT=$C; [ $C = default ] && unset GT_CW_CIT
  bash tools/gpu_r3.sh ${1:-q3}_$C prof 2>&1 | grep -E "convw" | cut -c1-120 | sed "s/^/cit=$C /"
done
