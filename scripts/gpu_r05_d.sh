#!/bin/bash
# round 5: dual chain with the heads' weight gradients + as1 mask at tile start (load-free dZ_as1 stage): parity, timeline, A/B
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python -m pytest tests/test_train_masks_gpu.py -m gpu -q > $OUT/r05d_masks.log 2>&1; tail -6 $OUT/r05d_masks.log
timeout 900 python -m pytest tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py -m gpu -q > $OUT/r05d_train_tests.log 2>&1; tail -4 $OUT/r05d_train_tests.log
L=$REPO/intrinsicnerf_amd
INERF_LIB_OVERRIDE=$L/libinerf_stamps.so python scripts/dgrad_timeline.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/r05d_timeline.txt
cat $OUT/r05d_timeline.txt
for form in dual single; do
  echo "== $form: $(INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E 'chain' )"
  echo "== $form coarse: $(INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E 'chain' )"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 2>&1 | grep "training step"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep "training step"
done > $OUT/r05d_ab.txt 2>&1
cat $OUT/r05d_ab.txt
for form in dual; do
  ( cd /tmp && INERF_DGRAD_KERNEL=$form rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r05d_train_$form -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
  find $OUT/prof/r05d_train_$form -name "*kernel_stats.csv" -exec cp {} $OUT/r05d_train_step_kernel_stats_$form.csv \;
  head -5 $OUT/r05d_train_step_kernel_stats_$form.csv | cut -c1-200
done
rm -rf $OUT/prof
