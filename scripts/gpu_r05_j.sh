#!/bin/bash
# round 5: amax pin in wide_store_h (inference kernels 251 -> 234 VGPRs, training forward 0 spills), dual chain without tid: full suite + same-box A/B
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/r05j_pytest_gpu.log 2>&1; tail -5 $OUT/r05j_pytest_gpu.log
L=$REPO/intrinsicnerf_amd
for rep in 1 2; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  echo "[$v $rep] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_mlp.py --rays 640000 --iters 3 --precision f16x3 2>&1 | tail -1)"
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "inference forward|training forward|gradient chain" | sed "s/^/[$v $rep] /"
done
done > $OUT/r05j_ab.txt 2>&1
cut -c1-170 $OUT/r05j_ab.txt
