#!/bin/bash
# Round 6, lease 7: MFMA shape / operand-statistics microbenchmark under the power cap
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/microbench/mfma_shapes.hip -o /tmp/mfma_shapes 2>/dev/null
timeout 600 /tmp/mfma_shapes 600000 > $OUT/r06_mfma_shapes.txt 2>&1; cat $OUT/r06_mfma_shapes.txt
