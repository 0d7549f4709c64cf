#!/bin/bash
# pp vs t128 under the weight-stream ablation (every fragment from one 4 KiB window: L1 hits) - is the pipelined form waiting for its weights?
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; L=$REPO/intrinsicnerf_amd
mkdir -p $OUT/prof; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
bash scripts/build_variant.sh abl_wl1 mlp_f16.hip,mlp_f16_t128.hip "-DINERF_ABL_WLOAD_L1" > /dev/null 2>&1
bash scripts/build_variant.sh abl_noepi mlp_f16.hip,mlp_f16_t128.hip "-DINERF_ABL_NO_EPILOGUE" > /dev/null 2>&1
export BENCH_SIZE="--rays 131072 --iters 2" BENCH_ARGS="--precision f16x3"
for v in base wl1; do
 for form in t128 pp; do
  lib=$L/libinerf.so; [ $v != base ] && lib=$L/libinerf_abl_$v.so
  export INERF_LIB_OVERRIDE=$lib INERF_F16_KERNEL=$form
  timeout -k 5 200 bash scripts/pmc_pass.sh ppa_${form}_$v GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU > /dev/null 2>&1
  python - $form $v <<'PY'
import csv, glob, sys, collections
form, v = sys.argv[1:3]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/prof/ppa_{form}_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_encode_mlp" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x[1:]) / max(1, len(x[1:])) for k, x in agg.items()}
cyc = m["GRBM_GUI_ACTIVE"] / 8
print(f"[{form:4s} {v:5s}] {cyc / 1e6:8.2f} Mcycles, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:5.1f} %, parked {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'] * 100:4.1f} % issue-stalled {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'] * 100:4.1f} %, VALU {m['SQ_INSTS_VALU']:.3e}")
PY
 done
done 2>&1 | tee $OUT/r06_pp_abl.txt
rm -rf $OUT/prof/ppa_*
