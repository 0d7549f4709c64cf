#!/bin/bash
# same-box A/B of the SSR MLP kernel: this tree against the previous commit (_ab_prev/), alternating
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in 1 2 3; do
  for tree in new prev; do
    dir=$REPO; [ $tree = prev ] && dir=$REPO/_ab_prev
    for c in 28 101; do
      ( cd $dir && timeout 300 python scripts/bench_mlp.py --ssr $c --rays 327680 --iters 3 2>&1 | tail -1 | sed "s/^/[$tree $rep C=$c] /" | cut -c1-170 )
    done
  done
done | tee gpurun_out/r04_ab_ssr_head.txt
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -q 2>&1 | tail -3
