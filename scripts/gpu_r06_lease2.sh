#!/bin/bash
# Round 6, lease 2: is the 128-point tile kernel bit-identical to the two-workgroup one, what does it do to the counters and the time
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 300 python scripts/diag_kernel_forms.py > $OUT/r06_t128_identity.txt 2>&1; tail -12 $OUT/r06_t128_identity.txt
for rep in 1 2; do for f in dual t128; do echo "[$f $rep] $(INERF_F16_KERNEL=$f python scripts/bench_mlp.py --rays 640000 --iters 3 --precision f16x3 2>&1 | tail -1)"; done; done > $OUT/r06_t128_ab.txt 2>&1
cat $OUT/r06_t128_ab.txt
timeout 700 bash scripts/gpu_r06_counters.sh dual > /dev/null 2>&1
timeout 700 bash scripts/gpu_r06_counters.sh t128 > /dev/null 2>&1
cat $OUT/r06_counters_dual.txt $OUT/r06_counters_t128.txt
timeout 600 python -m pytest tests/test_unfiltered_parity.py -m gpu -q -x -k "trained" 2>&1 | tail -15 > $OUT/r06_trained_tests.txt; cat $OUT/r06_trained_tests.txt
