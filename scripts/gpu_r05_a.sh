#!/bin/bash
# round 5, first lease: the new GPU tests, the bench line with the 32768-ray PSNR leg, plain `bench.py --gpus 2` (self-launch) on one GPU
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
timeout 600 python -m pytest tests/test_graphs_gpu.py tests/test_dropin_gpu.py -m gpu -q -x > $OUT/r05a_pytest.log 2>&1; tail -5 $OUT/r05a_pytest.log
( time timeout 900 python bench.py --steps 5 --warmup 1 ) > $OUT/r05a_bench.json 2> $OUT/r05a_bench.err; tail -c 600 $OUT/r05a_bench.err
( time INERF_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --cpu-baseline-quick ) > $OUT/r05a_bench_n2_self.json 2> $OUT/r05a_bench_n2_self.err; echo "n2 rc=$?"; tail -c 400 $OUT/r05a_bench_n2_self.err
python scripts/bench_train_kernels.py > $OUT/r05a_train_kernels.txt 2>&1; tail -8 $OUT/r05a_train_kernels.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a_bench.json"))
p = d["parity"]
print(d["value"], d["roofline"]["frac"], d["train_step"]["ms_per_step"], d["train_step"]["graphed_ms_per_step"], d["train_step"]["roofline"]["frac_eager"])
print({k: p[k] for k in p if k.startswith("psnr") and "per_map" not in k and "note" not in k and "fine_pass" not in k and "oracle" not in k})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
n2 = json.load(open("gpurun_out/r05a_bench_n2_self.json"))
print("n2:", n2["n_gpus"], n2["value"], n2["cpu_baseline"] is not None, n2["configs"]["ssr_room0_320x240"].get("checksum_identical_on_all_ranks"))
PY
