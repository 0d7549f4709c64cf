#!/bin/bash
# round 4, run n: shares of the four product shapes in the batched weight-gradient launch (development build with $INERF_WGRAD_WEIGHTS)
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
export INERF_LIB_OVERRIDE=$REPO/intrinsicnerf_amd/libinerf_tune.so
rm -f $OUT/r04n_shares.txt
for w in "40,28,32,18" "40,28,28,14" "40,28,24,10" "40,24,28,12" "40,32,28,12" "48,28,32,16" "40,20,24,10" "40,28,32,18"; do
  export INERF_WGRAD_WEIGHTS=$w
  rm -rf $OUT/prof/ab
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/ab -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > $OUT/prof/ab_step.txt 2>&1 )
  f=$(find $OUT/prof/ab -name "*kernel_stats.csv" | head -1)
  python - "$f" "$w" <<'PY' >> $OUT/r04n_shares.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = lambda sub: sum(float(r["TotalDurationNs"]) for r in rows if sub in r["Name"]) / 18 / 1e6
print(f"[{sys.argv[2]}] batched launch {tot('k_mlp_wgrad_frag'):.3f} ms per step, reduce {tot('k_reduce'):.3f}, all kernels {sum(float(r['TotalDurationNs']) for r in rows) / 18e6:.3f}")
PY
done
cat $OUT/r04n_shares.txt
rm -rf $OUT/prof
