#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02acc2}; mkdir -p $O
cd $R
for m in bf16x3 f32; do
  for sd in 1 2 3 4 5; do
    GT_PRECISION=$m timeout 300 python tools/accuracy_leg.py --impl hip --dropout-seed $sd 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$m', r['dropout_seed'], round(r['val_rel_l2'],4), round(r['train_loss_last'],4))" >> $O/acc.log
  done
done
cat $O/acc.log
