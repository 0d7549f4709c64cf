#!/bin/bash
# Same-box A/B: wide GEMMs whose first products take a zero C operand (wide_gemm_h PEEL; default) against explicit accumulator zeroing
# (-DINERF_T128_PEEL=0 -DINERF_DUAL_PEEL=0): object-level inference (128-point tile), SSR frame and training forward (two-workgroup kernel)
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; L=$REPO/intrinsicnerf_amd
mkdir -p $OUT; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
bash scripts/build_variant.sh nopeel mlp_f16_t128.hip,mlp_f16.hip "-DINERF_T128_PEEL=0 -DINERF_DUAL_PEEL=0" > /dev/null 2>&1
{
python scripts/diag_kernel_forms.py --forms dual,t128 --sizes 3x1,1x191,1000x192,4099x192 2>&1 | grep -v amdgpu | tail -3
for rep in 1 2 3; do
  for v in peel nopeel; do
    lib=$L/libinerf.so; [ $v = nopeel ] && lib=$L/libinerf_nopeel.so
    export INERF_LIB_OVERRIDE=$lib
    echo "[$v $rep t128 ] $(python scripts/bench_mlp.py --rays 262144 --iters 4 --precision f16x3 2>&1 | tail -1 | cut -c1-150)"
    echo "[$v $rep ssr  ] $(python scripts/bench_ssr_frame.py --frames 5 2>&1 | tail -1 | cut -c1-150)"
    echo "[$v $rep train] $(python scripts/bench_train_kernels.py 2>&1 | grep 'training forward')"
  done
done
} 2>&1 | tee $OUT/r06_peel_ab.txt
