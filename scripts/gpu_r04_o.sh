#!/bin/bash
# round 4, run o: chain stage variants against the previous commit
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_backward_golden.py tests/test_train_masks_gpu.py tests/test_trained_network_gpu.py tests/test_dropin_gpu.py tests/test_graphs_gpu.py -m gpu -x -q > $OUT/r04o_tests.txt 2>&1
tail -3 $OUT/r04o_tests.txt
rm -f $OUT/r04_ab_prof.txt
for rep in 1 2 3; do
  for tree in new prev; do
    dir=$REPO; [ $tree = prev ] && dir=$REPO/_ab_prev
    unset INERF_LIB_OVERRIDE; [ $tree = nopf ] && export INERF_LIB_OVERRIDE=$REPO/intrinsicnerf_amd/libinerf_nopf.so
    rm -rf $OUT/prof/ab
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/ab -o t -- python $dir/scripts/bench_train_step.py --iters 8 > $OUT/prof/ab_step.txt 2>&1 )
    f=$(find $OUT/prof/ab -name "*kernel_stats.csv" | head -1)
    python - "$f" "$tree" "$rep" "$(grep 'training step' $OUT/prof/ab_step.txt)" <<'PY' >> $OUT/r04_ab_prof.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def tot(sub):
    return sum(float(r["TotalDurationNs"]) for r in rows if sub in r["Name"]) / 18 / 1e6
all_ms = sum(float(r["TotalDurationNs"]) for r in rows) / 18 / 1e6
print(f"[{sys.argv[2]} {sys.argv[3]}] per step: forward {tot('k_encode_mlp'):.3f}  chain {tot('k_mlp_dgrad'):.3f}  wgrad batch {tot('k_mlp_wgrad_frag'):.3f}  "
      f"wgrad other {tot('k_mlp_wgrad<'):.3f}  reduce {tot('k_reduce'):.3f}  repack {tot('k_repack'):.3f}  all kernels {all_ms:.3f} ms | {sys.argv[4].strip()}")
PY
  done
done
unset INERF_LIB_OVERRIDE
cat $OUT/r04_ab_prof.txt
rm -rf $OUT/prof
