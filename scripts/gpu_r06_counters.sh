#!/bin/bash
# Round 6: what the headline kernel pulls through the CU's vector-memory path (L2 -> TCP -> registers), by counters.
#   bash scripts/gpu_r06_counters.sh [form]      (form: value of INERF_F16_KERNEL, default: the library's default)
# Writes gpurun_out/r06_counters_<form>.txt.  One counter group per pass, every pass under its own timeout (a counter name this
# rocprofv3 does not know makes it abort and then hang: lease 1 of the round lost 25 minutes to a TA_* pass).
form=${1:-default}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
[ "$form" != default ] && export INERF_F16_KERNEL=$form
python -c 'import __graft_entry__ as g; g.build()' || exit 1
export BENCH_SIZE="--rays 131072 --iters 2" BENCH_ARGS="--precision f16x3"
pass() { timeout -k 5 150 bash scripts/pmc_pass.sh "$@" || echo "pass $1: timed out / failed"; }
pass c6_${form}_sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pass c6_${form}_sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass c6_${form}_tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
pass c6_${form}_tcc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
{
  echo "# bench_mlp.py --rays 131072 --iters 2 --precision f16x3 (25 165 824 points), INERF_F16_KERNEL=$form; mean per dispatch of the k_encode_mlp launches"
  for p in sq sq2 tcp tcc; do python scripts/pmc_report.py $OUT/prof/c6_${form}_$p; done
  echo "# launch durations under the SQ pass (ns):"
  python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof/c6_${form}_sq/**/*kernel_trace.csv", recursive=True):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_encode_mlp" in r["Kernel_Name"]]
    print(" ", d)
PY
  echo "# un-profiled timing, same box:"
  python scripts/bench_mlp.py --rays 131072 --iters 5 --precision f16x3 2>&1 | tail -1
  python scripts/bench_mlp.py --rays 640000 --iters 3 --precision f16x3 2>&1 | tail -1
} > $OUT/r06_counters_${form}.txt 2>&1
cat $OUT/r06_counters_${form}.txt
rm -rf $OUT/prof
