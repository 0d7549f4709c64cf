#!/bin/bash
# round 5: fragment stores behind the consuming GEMM (training forward + dual chain): parity, then same-box A/B against the previous commit's kernels
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_train_masks_gpu.py tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py -m gpu -q -x > $OUT/r05h_tests.log 2>&1; tail -4 $OUT/r05h_tests.log
L=$REPO/intrinsicnerf_amd
for rep in 1 2; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "training forward|gradient chain|whole backward" | sed "s/^/[$v $rep] /"
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E "training forward|gradient chain" | sed "s/^/[$v $rep coarse] /"
done
done > $OUT/r05h_ab.txt 2>&1
cut -c1-140 $OUT/r05h_ab.txt
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  echo "[$v] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_step.py --iters 20 2>&1 | grep 'training step')"
  echo "[$v] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep 'training step')"
done
