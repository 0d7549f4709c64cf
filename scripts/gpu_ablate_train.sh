#!/bin/bash
# timing ablations of the training kernels (development builds of scripts/build_variant.sh; their RESULTS are wrong)
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd $REPO
for rep in 1 2; do
for v in "" f_nofragstore f_nofrag f_norows f_nosave b_nofragstore b_nofrag; do
  if [ -z "$v" ]; then unset INERF_LIB_OVERRIDE; tag=shipped; else export INERF_LIB_OVERRIDE=$REPO/intrinsicnerf_amd/libinerf_$v.so; tag=$v; fi
  timeout 300 python -W ignore scripts/bench_train_kernels.py 2>&1 | grep "inference forward\|training forward\|input-gradient" | awk -v t="[$tag $rep]" '{print t, $0}' | cut -c1-120
done
done | tee gpurun_out/r04_ablate_train.txt
