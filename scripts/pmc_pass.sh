#!/bin/bash
# usage: [BENCH_SCRIPT=scripts/bench_train_kernels.py BENCH_SIZE="--rays 2048 --iters 1"] scripts/pmc_pass.sh <tag> <counters...>
#        (run on the GPU box; writes gpurun_out/prof/<tag>/)
# Counter passes run separately from --stats/trace domains (gpurun refuses the combination).
tag=$1; shift
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $REPO/gpurun_out/prof
cd /tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/prof/$tag -o p -- python $REPO/${BENCH_SCRIPT:-scripts/bench_mlp.py} ${BENCH_SIZE:---rays 32768 --iters 2} $BENCH_ARGS > $REPO/gpurun_out/prof/$tag.log 2>&1
