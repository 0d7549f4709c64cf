#!/bin/bash
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python scripts/diag_fit_grads.py --steps 40 2>&1 | grep "vs fp64" | cut -c1-330 > $OUT/r04f_diag.txt; cat $OUT/r04f_diag.txt
( time timeout 900 python -m pytest tests/test_backward_golden.py tests/test_train_masks_gpu.py tests/test_graphs_gpu.py tests/test_dropin_gpu.py tests/test_range_fallback_gpu.py tests/test_trained_network_gpu.py tests/test_repack_gpu.py -m gpu -q --maxfail=10 ) > $OUT/r04f_pytest.log 2>&1
tail -12 $OUT/r04f_pytest.log
for rep in 1 2; do
  for tree in new old; do
    dir=$REPO; [ $tree = old ] && dir=$REPO/_ab_r03
    ( cd $dir && timeout 300 python scripts/bench_train_kernels.py 2>&1 | grep -v amdgpu.ids | grep -v library | sed "s/^/[$tree $rep] /" ) >> $OUT/r04f_ab_kernels.txt
    ( cd $dir && timeout 300 python scripts/bench_train_step.py --iters 10 2>&1 | grep "training step" | sed "s/^/[$tree $rep] /" ) >> $OUT/r04f_ab_step.txt
  done
done
cat $OUT/r04f_ab_kernels.txt $OUT/r04f_ab_step.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r04f_train_new -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
find $OUT/prof/r04f_train_new -name "*kernel_stats.csv" -exec cp {} $OUT/r04f_train_step_kernel_stats_new.csv \;
head -12 $OUT/r04f_train_step_kernel_stats_new.csv | cut -c1-150
rm -rf $OUT/prof
