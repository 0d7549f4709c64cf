#!/bin/bash
# round 5: full GPU suite + smoke on the two-workgroup chain / nt fragment stores; train step
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build(); g.smoke(); print("smoke ok")' > $OUT/r05g_smoke.txt 2>&1; tail -1 $OUT/r05g_smoke.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/r05g_pytest_gpu.log 2>&1; tail -6 $OUT/r05g_pytest_gpu.log
python scripts/bench_train_step.py --iters 20 2>&1 | grep "training step"
python scripts/bench_train_step.py --iters 20 --ssr 28 2>&1 | grep "training step"
