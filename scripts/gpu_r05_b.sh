#!/bin/bash
# round 5, second lease: the two-workgroup input-gradient chain - parity against the eight-wave chain and the training tests, then same-box A/B
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_train_masks_gpu.py -m gpu -q -x > $OUT/r05b_masks.log 2>&1; tail -15 $OUT/r05b_masks.log
timeout 900 python -m pytest tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py tests/test_range_fallback_gpu.py -m gpu -q > $OUT/r05b_train_tests.log 2>&1; tail -15 $OUT/r05b_train_tests.log
for form in single dual; do
  echo "== $form"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 9 2>&1 | grep -E "chain|whole backward" 
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 9 --samples 64 2>&1 | grep -E "chain|whole backward"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 2>&1 | grep -v amdgpu.ids | tail -3
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep -v amdgpu.ids | tail -1
done > $OUT/r05b_ab.txt 2>&1
cat $OUT/r05b_ab.txt
