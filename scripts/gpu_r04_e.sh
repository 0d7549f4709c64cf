#!/bin/bash
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python scripts/diag_fit_grads.py --steps 40 > gpurun_out/r04e_diag.txt 2>&1
cat gpurun_out/r04e_diag.txt | grep -v amdgpu.ids | cut -c1-400
( time timeout 900 python -m pytest tests/test_graphs_gpu.py tests/test_dropin_gpu.py tests/test_trained_network_gpu.py -m gpu -q --maxfail=10 ) > gpurun_out/r04e_pytest.log 2>&1
tail -15 gpurun_out/r04e_pytest.log
