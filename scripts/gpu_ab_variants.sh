#!/bin/bash
# same-box A/B of variant libraries (scripts/build_variant.sh) against the tree's: inference kernel (object, SSR) and training kernels
#   scripts/gpu_ab_variants.sh <tag> <variant>[,<variant>...] [reps]
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
L=$REPO/intrinsicnerf_amd
tag=$1; variants=${2//,/ }; reps=${3:-2}
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in $(seq $reps); do
  for v in base $variants; do
    lib=$L/libinerf.so; [ $v != base ] && lib=$L/libinerf_$v.so
    echo "[$v $rep] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_mlp.py --iters 7 2>&1 | grep k_encode_mlp | cut -c1-160)"
    echo "[$v $rep] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_mlp.py --iters 7 --ssr 28 --rays 32768 2>&1 | grep k_encode_mlp | cut -c1-160)"
    INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 7 2>&1 | grep "^training forward\|^input-gradient" | cut -c1-110 | sed "s/^/[$v $rep] /"
  done
done > $OUT/${tag}_ab.txt 2>&1
cat $OUT/${tag}_ab.txt
