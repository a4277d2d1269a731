#!/bin/bash
# GEMM ablation on the GPU box: same shapes, library variants with one phase removed.
for v in "" _nomfma _noload _nostore _onlymfma; do
  echo "== libgt_hip$v.so"
  GT_HIP_LIB=libgt_hip$v.so BATCH=${BATCH:-64} NOTORCH=1 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | cut -c1-110
done
