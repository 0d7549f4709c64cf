#!/bin/bash
# whole GPU suite + bench (N = 1), then the N = 2 shared-GPU run and its injected-failure path
mkdir -p gpurun_out
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 ) > gpurun_out/r04g_pytest.log 2>&1
tail -12 gpurun_out/r04g_pytest.log
( time timeout 1200 python bench.py --steps 5 --warmup 1 ) > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
tail -c 600 gpurun_out/r04g_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04g_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, "roofline", d["roofline"]["frac"], "strict", d["strict_fp32"], "train", d["train_step"]["ms_per_step"], d["train_step"]["graphed_ms_per_step"])
print("ssr", d["configs"]["ssr_room0_320x240"]["ms_per_step"], d["configs"]["ssr_room0_320x240"]["roofline"]["frac"], "parity", d["parity"]["violations"], d["parity"]["stagewise_violations"], d["parity"]["psnr_delta_db_rgb"], d["parity"]["psnr_delta_db_rgb_sampling_sigma"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
( time INERF_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-baseline-quick ) > gpurun_out/r04g_bench_n2_shared.json 2> gpurun_out/r04g_bench_n2_shared.err
echo "n2 rc=$?"; tail -c 400 gpurun_out/r04g_bench_n2_shared.err; python -c "
import json; d=json.loads(open('gpurun_out/r04g_bench_n2_shared.json').read().strip().splitlines()[-1]); print(d['n_gpus'], d['frame_costs']['per_rank'])"
( time INERF_BENCH_SHARE_GPU=1 INERF_BENCH_INJECT_FAILURE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-extras ) > gpurun_out/r04g_bench_n2_fail.json 2> gpurun_out/r04g_bench_n2_fail.err
echo "n2 injected failure rc=$? (must be non-zero)"; grep -c "parity violation\|does NOT match" gpurun_out/r04g_bench_n2_fail.err; tail -c 500 gpurun_out/r04g_bench_n2_fail.err
