#!/bin/bash
# same-box A/B of the training step's kernels: this tree against the round-3 tree (_ab_r03/), alternating, under rocprofv3
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in 1 2 3; do
  for tree in new old; do
    dir=$REPO; [ $tree = old ] && dir=$REPO/_ab_r03
    rm -rf $OUT/prof/ab
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/ab -o t -- python $dir/scripts/bench_train_step.py --iters 8 > $OUT/prof/ab_step.txt 2>&1 )
    f=$(find $OUT/prof/ab -name "*kernel_stats.csv" | head -1)
    python - "$f" "$tree" "$rep" "$(grep 'training step' $OUT/prof/ab_step.txt)" <<'PY' >> $OUT/r04_ab_prof.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def tot(sub, excl=None):
    return sum(float(r["TotalDurationNs"]) for r in rows if sub in r["Name"] and (excl is None or excl not in r["Name"])) / 18 / 1e6
all_ms = sum(float(r["TotalDurationNs"]) for r in rows) / 18 / 1e6
print(f"[{sys.argv[2]} {sys.argv[3]}] per step: forward {tot('k_encode_mlp'):.3f}  chain {tot('k_mlp_dgrad'):.3f}  wgrad256 {tot('k_mlp_wgrad_frag') + tot('k_mlp_wgrad_rows'):.3f}  "
      f"wgrad other {tot('k_mlp_wgrad<'):.3f}  reduce {tot('k_reduce'):.3f}  all kernels {all_ms:.3f} ms | {sys.argv[4].strip()}")
PY
  done
done
cat $OUT/r04_ab_prof.txt
rm -rf $OUT/prof
