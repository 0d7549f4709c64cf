#!/bin/bash
# round 4, third GPU call: same-box A/B of the training step against the round-3 tree (_ab_r03/), per-kernel times of both
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in 1 2; do
  for tree in new old; do
    dir=$REPO; [ $tree = old ] && dir=$REPO/_ab_r03
    ( cd $dir && timeout 300 python scripts/bench_train_kernels.py 2>&1 | grep -v amdgpu.ids | sed "s/^/[$tree $rep] /" ) >> $OUT/r04c_ab_kernels.txt
    ( cd $dir && timeout 300 python scripts/bench_train_step.py --iters 10 2>&1 | grep "training step" | sed "s/^/[$tree $rep] /" ) >> $OUT/r04c_ab_step.txt
  done
done
cat $OUT/r04c_ab_kernels.txt $OUT/r04c_ab_step.txt
for tree in new old; do
  dir=$REPO; [ $tree = old ] && dir=$REPO/_ab_r03
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r04c_train_$tree -o t -- python $dir/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
  find $OUT/prof/r04c_train_$tree -name "*kernel_stats.csv" -exec cp {} $OUT/r04c_train_step_kernel_stats_$tree.csv \;
  echo "== $tree"; head -14 $OUT/r04c_train_step_kernel_stats_$tree.csv | cut -c1-150
done
rm -rf $OUT/prof
