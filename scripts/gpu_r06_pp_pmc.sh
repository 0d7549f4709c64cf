#!/bin/bash
# Round 6: the pipelined trunk (INERF_F16_KERNEL=pp) against the 128-point tile it is built on, in cycles (PMC) and time, same box.
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
export BENCH_SIZE="--rays 131072 --iters 2" BENCH_ARGS="--precision f16x3"
{
for form in t128 pp; do
  export INERF_F16_KERNEL=$form
  timeout -k 5 200 bash scripts/pmc_pass.sh pp_${form}_a GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU > /dev/null 2>&1
  timeout -k 5 200 bash scripts/pmc_pass.sh pp_${form}_b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM > /dev/null 2>&1
  python - $form <<'PY'
import csv, glob, sys, collections
form = sys.argv[1]
m = {}
for tag in ("a", "b"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/prof/pp_{form}_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_encode_mlp" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m.update({k: sum(x[1:]) / max(1, len(x[1:])) for k, x in agg.items()})
cyc = m["GRBM_GUI_ACTIVE"] / 8
print(f"[{form:4s}] {cyc / 1e6:8.2f} Mcycles per launch, MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:5.1f} %, wave cycles: parked {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'] * 100:4.1f} % "
      f"issue-stalled {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'] * 100:4.1f} %, VALU insts {m['SQ_INSTS_VALU']:.3e}, LDS insts {m.get('SQ_INSTS_LDS', 0):.3e} conflicts {m.get('SQ_LDS_BANK_CONFLICT', 0):.3e} "
      f"idx-active {m.get('SQ_LDS_IDX_ACTIVE', 0):.3e}, wait-LDS {m.get('SQ_WAIT_INST_LDS', 0):.3e}, VMEM rd insts {m.get('SQ_INSTS_VMEM_RD', 0):.3e}, SALU {m.get('SQ_INSTS_SALU', 0):.3e}")
PY
  echo "   un-profiled: $(python scripts/bench_mlp.py --rays 131072 --iters 4 --precision f16x3 2>&1 | tail -1 | sed 's/.*median \([0-9.]*\) ms.*-> \([0-9.]*\) TFLOP.*/\1 ms \2 TFLOP\/s/')"
done
} 2>&1 | tee $OUT/r06_pp_pmc.txt
rm -rf $OUT/prof/pp_*
