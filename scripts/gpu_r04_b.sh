#!/bin/bash
# round 4, second GPU call: whole GPU suite, kernel / step timings, bench with the N = 2 shared-GPU failure path
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 ) > gpurun_out/r04b_pytest.log 2>&1
tail -45 gpurun_out/r04b_pytest.log
timeout 300 python scripts/bench_train_kernels.py > gpurun_out/r04b_train_kernels.txt 2>&1
tail -8 gpurun_out/r04b_train_kernels.txt
timeout 300 python scripts/bench_train_step.py --iters 10 > gpurun_out/r04b_train_step.txt 2>&1
timeout 300 python scripts/bench_train_step.py --iters 10 --ssr 28 >> gpurun_out/r04b_train_step.txt 2>&1
grep "training step" gpurun_out/r04b_train_step.txt
