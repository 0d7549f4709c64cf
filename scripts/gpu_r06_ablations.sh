#!/bin/bash
# Round 6: where do the headline kernel's cycles go?  Timing ablations (development builds whose RESULTS ARE WRONG: scripts/build_variant.sh
# abl_<x> mlp_f16.hip,mlp_f16_t128.hip -DINERF_ABL_<X>) of the 128-point and the 64-point form, compared in CYCLES (GRBM_GUI_ACTIVE / 8: the
# planes' contents and with them the power, hence the clock, differ between the builds) and in time.
#   nobar: no layer barriers | wl1: every weight fragment from one 4 KiB window (same instructions, no L2 -> CU stream) | noenc: first tile's
#   encoding only | noepi: no epilogue (bias, ReLU, split, LDS stores) | all: the four together (what is left: the GEMM loops and the heads)
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
L=$REPO/intrinsicnerf_amd
mkdir -p $OUT/prof
cd $REPO
export BENCH_SIZE="--rays 131072 --iters 2" BENCH_ARGS="--precision f16x3"
{
for form in t128 dual; do
  for v in base nobar wl1 noenc noepi all; do
    lib=$L/libinerf.so; [ $v != base ] && lib=$L/libinerf_abl_$v.so
    export INERF_LIB_OVERRIDE=$lib INERF_F16_KERNEL=$form
    timeout -k 5 200 bash scripts/pmc_pass.sh ab_${form}_$v GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU > /dev/null 2>&1
    python - $form $v <<'PY'
import csv, glob, sys, collections
form, v = sys.argv[1:3]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/prof/ab_{form}_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_encode_mlp" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x[1:]) / max(1, len(x[1:])) for k, x in agg.items()}
if m:
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print(f"[{form:4s} {v:5s}] {cyc / 1e6:8.2f} Mcycles per launch, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:5.1f} %, wave cycles: parked {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'] * 100:4.1f} % issue-stalled {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'] * 100:4.1f} %, VALU insts {m['SQ_INSTS_VALU']:.3e}", end="")
else:
    print(f"[{form} {v}] no counters", end="")
PY
    echo " | un-profiled: $(python scripts/bench_mlp.py --rays 131072 --iters 4 --precision f16x3 2>&1 | tail -1 | sed 's/.*median \([0-9.]*\) ms.*-> \([0-9.]*\) TFLOP.*/\1 ms \2 TFLOP\/s/')"
  done
done
} > $OUT/r06_ablations.txt 2>&1
cat $OUT/r06_ablations.txt
rm -rf $OUT/prof
