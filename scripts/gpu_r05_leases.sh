#!/bin/bash
# The GPU leases of round 5, one function per gpurun call, in the order they ran (development record; `bash scripts/gpu_r05_leases.sh <name>`).
# Variant libraries (libinerf_<name>.so) come from scripts/build_variant.sh in the build container before the call.
export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; L=$REPO/intrinsicnerf_amd; cd $REPO

lease_a() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python -m pytest tests/test_graphs_gpu.py tests/test_dropin_gpu.py -m gpu -q -x > $OUT/r05a_pytest.log 2>&1; tail -5 $OUT/r05a_pytest.log
( time timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/r05a_bench.json 2> $OUT/r05a_bench.err; tail -c 600 $OUT/r05a_bench.err
( time INERF_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --cpu-baseline-quick ) > $OUT/r05a_bench_n2_self.json 2> $OUT/r05a_bench_n2_self.err; echo "n2 rc=$?"; tail -c 400 $OUT/r05a_bench_n2_self.err
python scripts/bench_train_kernels.py > $OUT/r05a_train_kernels.txt 2>&1; tail -8 $OUT/r05a_train_kernels.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a_bench.json"))
p = d["parity"]
print(d["value"], d["roofline"]["frac"], d["train_step"]["ms_per_step"], d["train_step"]["graphed_ms_per_step"], d["train_step"]["roofline"]["frac_eager"])
print({k: p[k] for k in p if k.startswith("psnr") and "per_map" not in k and "note" not in k and "fine_pass" not in k and "oracle" not in k})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
n2 = json.load(open("gpurun_out/r05a_bench_n2_self.json"))
print("n2:", n2["n_gpus"], n2["value"], n2["cpu_baseline"] is not None, n2["configs"]["ssr_room0_320x240"].get("checksum_identical_on_all_ranks"))
PY
}

lease_b() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_train_masks_gpu.py -m gpu -q -x > $OUT/r05b_masks.log 2>&1; tail -15 $OUT/r05b_masks.log
timeout 900 python -m pytest tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py tests/test_range_fallback_gpu.py -m gpu -q > $OUT/r05b_train_tests.log 2>&1; tail -15 $OUT/r05b_train_tests.log
for form in single dual; do
  echo "== $form"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 9 2>&1 | grep -E "chain|whole backward" 
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 9 --samples 64 2>&1 | grep -E "chain|whole backward"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 2>&1 | grep -v amdgpu.ids | tail -3
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep -v amdgpu.ids | tail -1
done > $OUT/r05b_ab.txt 2>&1
cat $OUT/r05b_ab.txt
}

lease_c() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python -m pytest tests/test_train_masks_gpu.py -m gpu -q > $OUT/r05c_masks.log 2>&1; tail -6 $OUT/r05c_masks.log
( INERF_LIB_OVERRIDE=$L/libinerf_stamps.so python scripts/dgrad_timeline.py; INERF_DGRAD_KERNEL=single INERF_LIB_OVERRIDE=$L/libinerf_stamps.so python scripts/dgrad_timeline.py ) 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/r05c_timeline.txt
cat $OUT/r05c_timeline.txt
for v in base pipe nostag; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  echo "== $v: $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E 'chain' )"
  echo "== $v coarse: $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E 'chain' )"
done > $OUT/r05c_variants.txt 2>&1
echo "== single: $(INERF_DGRAD_KERNEL=single python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E 'chain' )" >> $OUT/r05c_variants.txt
cat $OUT/r05c_variants.txt
for form in dual single; do
  ( cd /tmp && INERF_DGRAD_KERNEL=$form rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r05c_train_$form -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
  find $OUT/prof/r05c_train_$form -name "*kernel_stats.csv" -exec cp {} $OUT/r05c_train_step_kernel_stats_$form.csv \;
  head -8 $OUT/r05c_train_step_kernel_stats_$form.csv | cut -c1-200
done
rm -rf $OUT/prof
}

lease_d() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python -m pytest tests/test_train_masks_gpu.py -m gpu -q > $OUT/r05d_masks.log 2>&1; tail -6 $OUT/r05d_masks.log
timeout 900 python -m pytest tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py -m gpu -q > $OUT/r05d_train_tests.log 2>&1; tail -4 $OUT/r05d_train_tests.log
INERF_LIB_OVERRIDE=$L/libinerf_stamps.so python scripts/dgrad_timeline.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/r05d_timeline.txt
cat $OUT/r05d_timeline.txt
for form in dual single; do
  echo "== $form: $(INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E 'chain' )"
  echo "== $form coarse: $(INERF_DGRAD_KERNEL=$form python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E 'chain' )"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 2>&1 | grep "training step"
  INERF_DGRAD_KERNEL=$form python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep "training step"
done > $OUT/r05d_ab.txt 2>&1
cat $OUT/r05d_ab.txt
for form in dual; do
  ( cd /tmp && INERF_DGRAD_KERNEL=$form rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r05d_train_$form -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
  find $OUT/prof/r05d_train_$form -name "*kernel_stats.csv" -exec cp {} $OUT/r05d_train_step_kernel_stats_$form.csv \;
  head -5 $OUT/r05d_train_step_kernel_stats_$form.csv | cut -c1-200
done
rm -rf $OUT/prof
}

lease_e() {
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
}

lease_f() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for rep in 1 2; do
for v in base aux2 aux16 aux18 aux1 nostore; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "training forward|gradient chain|whole backward" | sed "s/^/[$v $rep] /"
done
done > $OUT/r05f_aux.txt 2>&1
cat $OUT/r05f_aux.txt | cut -c1-150
}

lease_g() {
python -c 'import __graft_entry__ as g; g.build(); g.smoke(); print("smoke ok")' > $OUT/r05g_smoke.txt 2>&1; tail -1 $OUT/r05g_smoke.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/r05g_pytest_gpu.log 2>&1; tail -6 $OUT/r05g_pytest_gpu.log
python scripts/bench_train_step.py --iters 20 2>&1 | grep "training step"
python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep "training step"
}

lease_h() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_train_masks_gpu.py tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py -m gpu -q -x > $OUT/r05h_tests.log 2>&1; tail -4 $OUT/r05h_tests.log
for rep in 1 2; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "training forward|gradient chain|whole backward" | sed "s/^/[$v $rep] /"
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E "training forward|gradient chain" | sed "s/^/[$v $rep coarse] /"
done
done > $OUT/r05h_ab.txt 2>&1
cut -c1-140 $OUT/r05h_ab.txt
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  echo "[$v] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_step.py --iters 20 2>&1 | grep 'training step')"
  echo "[$v] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep 'training step')"
done
}

lease_i() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
python scripts/diag_chain_forms.py 2>&1 | grep -v amdgpu.ids > $OUT/r05i_diag.txt; cat $OUT/r05i_diag.txt
timeout 600 python -m pytest tests/test_train_masks_gpu.py -m gpu -q > $OUT/r05i_masks.log 2>&1; tail -8 $OUT/r05i_masks.log
}

lease_j() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/r05j_pytest_gpu.log 2>&1; tail -5 $OUT/r05j_pytest_gpu.log
for rep in 1 2; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  echo "[$v $rep] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_mlp.py --rays 640000 --iters 3 --precision f16x3 2>&1 | tail -1)"
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "inference forward|training forward|gradient chain" | sed "s/^/[$v $rep] /"
done
done > $OUT/r05j_ab.txt 2>&1
cut -c1-170 $OUT/r05j_ab.txt
}

lease_k() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_train_masks_gpu.py tests/test_backward_golden.py tests/test_graphs_gpu.py tests/test_trained_network_gpu.py -m gpu -q > $OUT/r05k_tests.log 2>&1; tail -4 $OUT/r05k_tests.log
for rep in 1 2 3; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 2>&1 | grep -E "gradient chain" | sed "s/^/[$v $rep] /"
  INERF_LIB_OVERRIDE=$lib python scripts/bench_train_kernels.py --iters 15 --samples 64 2>&1 | grep -E "gradient chain" | sed "s/^/[$v $rep coarse] /"
done
done > $OUT/r05k_ab.txt 2>&1
cut -c1-140 $OUT/r05k_ab.txt
}

lease_l() {
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_unfiltered_parity.py tests/test_wrappers_gpu.py -m gpu -q -k "ssr or SSR or sem or csplit or split" > $OUT/r05l_tests.log 2>&1; tail -3 $OUT/r05l_tests.log
for rep in 1 2 3; do
for v in base prev; do
  lib=$L/libinerf_$v.so; [ $v = base ] && lib=$L/libinerf.so
  for c in 28 101; do
    echo "[$v $rep C=$c] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_ssr_frame.py --frames 6 --classes $c 2>&1 | grep -v amdgpu | tail -1)"
  done
done
done > $OUT/r05l_ab.txt 2>&1
cut -c1-220 $OUT/r05l_ab.txt
}

"lease_$1"
