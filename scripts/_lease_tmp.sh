export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
python scripts/diag_chain_repeat.py --launches 1500 2>&1 | grep -v "amdgpu.ids" | tail -3 | cut -c1-300
python scripts/diag_chain_repeat.py --launches 300 --rays 2048 --samples 192 2>&1 | grep -v "amdgpu.ids" | tail -3 | cut -c1-300
INERF_DGRAD_KERNEL=single python scripts/diag_chain_repeat.py --launches 600 2>&1 | grep -v "amdgpu.ids" | tail -1 | cut -c1-300
timeout 900 python -m pytest tests/test_train_masks_gpu.py tests/test_backward_golden.py -m gpu -q 2>&1 | tail -2
