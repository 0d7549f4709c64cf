#!/bin/bash
# round 4, run p: SSR semantic hidden layer's gradient as a fragment slot (its product joins the batched launch)
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_backward_golden.py tests/test_train_masks_gpu.py tests/test_dropin_gpu.py tests/test_graphs_gpu.py -m gpu -x -q > $OUT/r04p_tests.txt 2>&1
tail -3 $OUT/r04p_tests.txt
rm -f $OUT/r04p_ssr_step.txt
for rep in 1; do
  for tree in new prev; do
    dir=$REPO; [ $tree = prev ] && dir=$REPO/_ab_prev
    echo "[$tree $rep] $(python $dir/scripts/bench_train_step.py --iters 10 --ssr 28 2>&1 | grep 'training step')" >> $OUT/r04p_ssr_step.txt
  done
done
cat $OUT/r04p_ssr_step.txt
