#!/bin/bash
# Round 6, lease 6: launch-size sweep; the GPU suite on the final kernels
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( python scripts/bench_launch_size.py; INERF_F16_KERNEL=dual python scripts/bench_launch_size.py; python scripts/bench_launch_size.py --ssr 28; python scripts/bench_launch_size.py --samples 64 ) 2>&1 | grep -v amdgpu > $OUT/r06_launch_size.txt; cat $OUT/r06_launch_size.txt
( time timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -15 ) > $OUT/r06_l6_pytest_gpu.txt 2>&1; tail -6 $OUT/r06_l6_pytest_gpu.txt
