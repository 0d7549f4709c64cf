#!/bin/bash
# round 4, first GPU call: the fragment-slot training path - focused tests, then kernel and step timings
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 900 python -m pytest tests/test_backward_golden.py tests/test_train_masks_gpu.py -m gpu -q -x --maxfail=8 \
    -k "fragment or one_call or masks or repeats or network_backward_vs_torch or training_step_gradients or splits_large or weight_gradient_kernel" ) > gpurun_out/r04a_pytest.log 2>&1
tail -30 gpurun_out/r04a_pytest.log
timeout 300 python scripts/bench_train_kernels.py > gpurun_out/r04a_train_kernels.txt 2>&1
cat gpurun_out/r04a_train_kernels.txt | tail -12
timeout 300 python scripts/bench_train_step.py --iters 10 > gpurun_out/r04a_train_step.txt 2>&1
tail -6 gpurun_out/r04a_train_step.txt
timeout 200 python scripts/bench_mlp.py --rays 327680 --iters 3 > gpurun_out/r04a_bench_mlp.txt 2>&1
tail -3 gpurun_out/r04a_bench_mlp.txt
