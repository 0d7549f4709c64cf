#!/bin/bash
# Round 6: do the two co-resident workgroups of k_encode_mlp_f16x3_dual run in lockstep?  Development builds in which one workgroup of each
# pair (mode 1: the upper half of the grid; mode 2: odd workgroups) starts 2 / 3 / 5 x 2048 cycles late; same-box, object-level inference.
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out; L=$REPO/intrinsicnerf_amd
mkdir -p $OUT; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
for m in 1 2; do for u in 2 3 5; do bash scripts/build_variant.sh dph${m}_$u mlp_f16.hip "-DINERF_DEPHASE=$m -DINERF_DEPHASE_UNITS=$u" > /dev/null 2>&1; done; done
{
for rep in 1 2; do
  for v in base dph1_2 dph1_3 dph1_5 dph2_2 dph2_3 dph2_5; do
    lib=$L/libinerf.so; [ $v != base ] && lib=$L/libinerf_$v.so
    echo "[$v $rep dual] $(INERF_LIB_OVERRIDE=$lib INERF_F16_KERNEL=dual python scripts/bench_mlp.py --rays 262144 --iters 4 --precision f16x3 2>&1 | tail -1)"
  done
done
} | tee $OUT/r06_dephase.txt
