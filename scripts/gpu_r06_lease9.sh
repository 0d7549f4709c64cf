#!/bin/bash
# Round 6, lease 9: v_pk_maximum3_f16 in the epilogues' running maximum, same-box A/B (base = with, nomax3 = two v_pk_max_f16), forms' identity
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
L=$REPO/intrinsicnerf_amd
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 300 python scripts/diag_kernel_forms.py --sizes 3x1,1x191,33x64,4099x192 2>&1 | grep -v amdgpu | tail -6
for rep in 1 2 3; do for v in base nomax3; do
  lib=$L/libinerf.so; [ $v != base ] && lib=$L/libinerf_$v.so
  echo "[$v $rep t128] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_mlp.py --rays 262144 --iters 3 --precision f16x3 2>&1 | tail -1 | cut -c1-170)"
  echo "[$v $rep ssr ] $(INERF_LIB_OVERRIDE=$lib python scripts/bench_mlp.py --rays 131072 --iters 3 --ssr 28 --precision f16x3 2>&1 | tail -1 | cut -c1-170)"
done; done > $OUT/r06_max3_ab.txt 2>&1
cat $OUT/r06_max3_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_range_fallback_gpu.py -m gpu -q -x 2>&1 | tail -3
