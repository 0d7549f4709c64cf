#!/bin/bash
# round 5: cache policy of the 1 KB fragment stores (training forward + chain), same box
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
L=$REPO/intrinsicnerf_amd
for rep in 1 2; do
for v in base aux2 aux16 aux18 aux1 nostore; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "training forward|gradient chain|whole backward" | sed "s/^/[$v $rep] /"
done
done > $OUT/r05f_aux.txt 2>&1
cat $OUT/r05f_aux.txt | cut -c1-150
