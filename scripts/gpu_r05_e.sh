#!/bin/bash
# round 5: what does the dual chain wait for?  PMC passes (instruction cache, wait states, VMEM / LDS issue) on both chain forms
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|SQC|TCP|TA|TCC)_[A-Z0-9_]+" | sort -u > $OUT/r05e_counters.txt ); wc -l $OUT/r05e_counters.txt
grep -E "ICACHE|IFETCH|WAIT|STALL|BUSY" $OUT/r05e_counters.txt | tr '\n' ' '
export BENCH_SCRIPT=scripts/bench_train_kernels.py BENCH_SIZE="--rays 2048 --iters 1" BENCH_ARGS=""
for form in dual single; do
  export INERF_DGRAD_KERNEL=$form
  bash scripts/pmc_pass.sh r05e_${form}_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
  bash scripts/pmc_pass.sh r05e_${form}_b SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
  bash scripts/pmc_pass.sh r05e_${form}_c SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
  for p in a b c; do f=$(find $OUT/prof/r05e_${form}_$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "dgrad" in k or "k_encode_mlp_f16x3_dual<true" in k:
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: f"{max(v):.4g}" for c, v in d.items()})
PY
  done
done > $OUT/r05e_pmc.txt 2>&1
cat $OUT/r05e_pmc.txt
tail -3 $OUT/prof/r05e_dual_b.log
rm -rf $OUT/prof
