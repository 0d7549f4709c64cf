#!/bin/bash
# same-box A/B of the batched weight-gradient launch: this tree's library against intrinsicnerf_amd/libinerf_prev.so
# (scripts/build_variant.sh prev mlp_wgrad.hip ""), alternating, under rocprofv3; then the bit-repeat / golden tests on the new one
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
L=$REPO/intrinsicnerf_amd
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in 1 2; do
  for v in new prev; do
    lib=$L/libinerf.so; [ $v = prev ] && lib=$L/libinerf_prev.so
    rm -rf $OUT/prof/ab
    ( cd /tmp && INERF_LIB_OVERRIDE=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/ab -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > $OUT/prof/ab_step.txt 2>&1 )
    f=$(find $OUT/prof/ab -name "*kernel_stats.csv" | head -1)
    python - "$f" "$v" "$rep" "$(grep 'training step' $OUT/prof/ab_step.txt)" <<'PY' >> $OUT/r05_ab_wgrad.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def one(sub):
    r = [r for r in rows if sub in r["Name"]]
    return sum(float(x["TotalDurationNs"]) for x in r) / max(1, sum(int(x["Calls"]) for x in r)) / 1e3, sum(int(x["Calls"]) for x in r)
a, na = one("k_mlp_wgrad_frag"); b, nb = one("k_mlp_wgrad_rows"); c, nc = one("k_mlp_dgrad")
print(f"[{sys.argv[2]} {sys.argv[3]}] wgrad_frag {a:.1f} us x{na}  wgrad_rows {b:.1f} us x{nb}  chain {c:.1f} us x{nc} | {sys.argv[4].strip()}")
PY
  done
done
cat $OUT/r05_ab_wgrad.txt
rm -rf $OUT/prof
timeout 600 python -m pytest tests/test_backward_golden.py tests/test_train_masks_gpu.py tests/test_graphs_gpu.py -m gpu -q -x 2>&1 | tail -3
