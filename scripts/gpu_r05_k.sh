#!/bin/bash
# round 5: head-gradient loads first + block-wise |dz| bound in the chain epilogues: parity + same-box A/B
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_train_masks_gpu.py tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py -m gpu -q > $OUT/r05k_tests.log 2>&1; tail -4 $OUT/r05k_tests.log
L=$REPO/intrinsicnerf_amd
for rep in 1 2 3; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "gradient chain" | sed "s/^/[$v $rep] /"
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E "gradient chain" | sed "s/^/[$v $rep coarse] /"
done
done > $OUT/r05k_ab.txt 2>&1
cut -c1-140 $OUT/r05k_ab.txt
