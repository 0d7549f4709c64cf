#!/bin/bash
# Round 6, lease 1: what the headline kernel pulls through the CU's vector-memory path (L2 -> TCP -> registers), by counters.
#   bash scripts/gpu_r06_counters.sh [form]      (form: value of INERF_F16_KERNEL, default: the library's default)
# Writes gpurun_out/r06_counters_<form>.txt (+ the list of counters this rocprofv3 knows).  One counter group per pass.
form=${1:-default}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT/prof
cd $REPO
[ "$form" != default ] && export INERF_F16_KERNEL=$form
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( cd /tmp && rocprofv3 -L > $OUT/r06_rocprofv3_counters.txt 2>&1 )
grep -c . $OUT/r06_rocprofv3_counters.txt
export BENCH_SIZE="--rays 131072 --iters 2" BENCH_ARGS="--precision f16x3"
bash scripts/pmc_pass.sh c6_${form}_sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
bash scripts/pmc_pass.sh c6_${form}_sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
bash scripts/pmc_pass.sh c6_${form}_tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
bash scripts/pmc_pass.sh c6_${form}_tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
bash scripts/pmc_pass.sh c6_${form}_ta TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
bash scripts/pmc_pass.sh c6_${form}_td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum
bash scripts/pmc_pass.sh c6_${form}_tcc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
bash scripts/pmc_pass.sh c6_${form}_tcc2 TCC_BUSY_sum TCC_TAG_STALL_sum TCC_CYCLE_sum
{
  echo "# bench_mlp.py --rays 131072 --iters 2 --precision f16x3, INERF_F16_KERNEL=$form; mean per dispatch of the k_encode_mlp launches"
  for p in sq sq2 tcp tcp2 ta td tcc tcc2; do python scripts/pmc_report.py $OUT/prof/c6_${form}_$p; grep -i "error\|invalid\|not found" $OUT/prof/c6_${form}_$p.log | head -3; done
  echo "# un-profiled timing, same box:"
  python scripts/bench_mlp.py --rays 131072 --iters 5 --precision f16x3 2>&1 | tail -1
  python scripts/bench_mlp.py --rays 640000 --iters 3 --precision f16x3 2>&1 | tail -1
} > $OUT/r06_counters_${form}.txt 2>&1
cat $OUT/r06_counters_${form}.txt
rm -rf $OUT/prof
