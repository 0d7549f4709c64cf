#!/bin/bash
# One gpurun call of the round's standing checks: GPU test-suite, bench at N = 1, bench at N = 2 sharing the one GPU (gloo).
# usage (from the repo root, on the GPU box):  bash scripts/gpu_round.sh <tag> [pytest args...]
tag=${1:-r03}; shift
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1100 python -m pytest tests -m gpu -q --maxfail=25 "$@" ) > gpurun_out/${tag}_pytest.log 2>&1
tail -40 gpurun_out/${tag}_pytest.log
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench.err
INERF_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${tag}_bench_n2_shared.json 2> gpurun_out/${tag}_bench_n2_shared.err
tail -c 1500 gpurun_out/${tag}_bench_n2_shared.err
