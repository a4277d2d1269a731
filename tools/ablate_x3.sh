#!/bin/bash
# In-situ ablation of the split-operand ring kernel: the whole training step with library variants in which one phase
# of gemm_x3r_kernel is removed (results are wrong; only ms/step is read).  python galerkin-transformer_amd/build.py --ablate-x3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02abl}; mkdir -p $O
cd $R
for v in "" _x3nomfma _x3nosplit _x3nosplitb _x3nostore _x3onlyload; do
  echo -n "libgt_hip$v.so  " >> $O/ablate.log
  GT_HIP_LIB=libgt_hip$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])" >> $O/ablate.log
done
for p in ; do
  echo -n "precision $p  " >> $O/ablate.log
  GT_PRECISION=$p timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])" >> $O/ablate.log
done
cat $O/ablate.log
