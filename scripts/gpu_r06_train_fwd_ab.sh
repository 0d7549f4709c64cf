#!/bin/bash
# Same-box A/B of the training forward's tile forms (64-point two-workgroup kernel vs INERF_TRAIN_FWD=t128), kernel alone and whole step.
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out
mkdir -p $OUT
python -c 'import __graft_entry__ as g; g.build()' || exit 1
{
for rep in 1 2 3; do
  for form in dual t128; do
    if [ $form = t128 ]; then export INERF_TRAIN_FWD=t128; else unset INERF_TRAIN_FWD; fi
    echo "[$form $rep] $(python scripts/bench_train_kernels.py 2>&1 | grep 'training forward')"
    echo "[$form $rep] $(python scripts/bench_train_step.py --iters 8 2>&1 | tail -1)"
  done
done
} | tee $OUT/r06_train_fwd_ab.txt
