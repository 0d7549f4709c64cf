#!/bin/bash
# round 5, third lease: dual chain - bit equality with the eight-wave chain, phase timeline, variants (same box), rocprof of the step
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python -m pytest tests/test_train_masks_gpu.py -m gpu -q > $OUT/r05c_masks.log 2>&1; tail -6 $OUT/r05c_masks.log
L=$REPO/intrinsicnerf_amd
( INERF_LIB_OVERRIDE=$L/libinerf_stamps.so python scripts/dgrad_timeline.py; INERF_DGRAD_KERNEL=single INERF_LIB_OVERRIDE=$L/libinerf_stamps.so python scripts/dgrad_timeline.py ) 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/r05c_timeline.txt
cat $OUT/r05c_timeline.txt
for v in base pipe nostag; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  echo "== $v: $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E 'chain' )"
  echo "== $v coarse: $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E 'chain' )"
done > $OUT/r05c_variants.txt 2>&1
echo "== single: $(INERF_DGRAD_KERNEL=single python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E 'chain' )" >> $OUT/r05c_variants.txt
cat $OUT/r05c_variants.txt
for form in dual single; do
  ( cd /tmp && INERF_DGRAD_KERNEL=$form rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r05c_train_$form -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
  find $OUT/prof/r05c_train_$form -name "*kernel_stats.csv" -exec cp {} $OUT/r05c_train_step_kernel_stats_$form.csv \;
  head -8 $OUT/r05c_train_step_kernel_stats_$form.csv | cut -c1-200
done
rm -rf $OUT/prof
