#!/bin/bash
# Same-box A/B: the epilogues' bias/scale fma as v_pk_fma_f32 (two values per instruction; -DINERF_EPI_PKFMA=1) against v_fma_f32
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; L=$REPO/intrinsicnerf_amd
mkdir -p $OUT; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
bash scripts/build_variant.sh pkfma mlp_f16_t128.hip,mlp_f16.hip "-DINERF_EPI_PKFMA=1" > /dev/null 2>&1
{
INERF_LIB_OVERRIDE=$L/libinerf_pkfma.so python scripts/bench_mlp.py --rays 4099 --iters 2 --precision f16x3 2>&1 | tail -1 | cut -c1-200
python scripts/bench_mlp.py --rays 4099 --iters 2 --precision f16x3 2>&1 | tail -1 | cut -c1-200
for rep in 1 2 3; do
  for v in base pkfma; do
    lib=$L/libinerf.so; [ $v = pkfma ] && lib=$L/libinerf_pkfma.so
    export INERF_LIB_OVERRIDE=$lib
    echo "[$v $rep t128 ] $(python scripts/bench_mlp.py --rays 262144 --iters 4 --precision f16x3 2>&1 | tail -1 | cut -c1-170)"
    echo "[$v $rep ssr  ] $(python scripts/bench_ssr_frame.py --frames 5 2>&1 | tail -1 | cut -c1-150)"
  done
done
} 2>&1 | tee $OUT/r06_pkfma_ab.txt
