#!/bin/bash
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
python scripts/diag_chain_forms.py 2>&1 | grep -v amdgpu.ids > $OUT/r05i_diag.txt; cat $OUT/r05i_diag.txt
timeout 600 python -m pytest tests/test_train_masks_gpu.py -m gpu -q > $OUT/r05i_masks.log 2>&1; tail -8 $OUT/r05i_masks.log
