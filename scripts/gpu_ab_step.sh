#!/bin/bash
# same-box A/B of the whole training step: the tree's library against variant libraries, alternating
#   scripts/gpu_ab_step.sh <tag> <variant>[,<variant>...] [reps]
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
L=$REPO/intrinsicnerf_amd
tag=$1; variants=${2//,/ }; reps=${3:-3}
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in $(seq $reps); do
  for v in base $variants; do
    lib=$L/libinerf.so; [ $v != base ] && lib=$L/libinerf_$v.so
    echo "[$v $rep] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_step.py --iters 30 2>&1 | grep 'training step' | cut -c1-120)"
  done
done > $OUT/${tag}_ab.txt 2>&1
cat $OUT/${tag}_ab.txt
