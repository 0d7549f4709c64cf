#!/bin/bash
# Last lease of a round: smoke(), the GPU suite, the bench line, the headline kernel's HBM traffic and the training steps on the FINAL code
#   bash scripts/gpu_final.sh r05      (writes gpurun_out/<tag>_final_*; FINAL_SHORT=1: smoke, suite and bench only)
tag=${1:-rXX}
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > $OUT/${tag}_final_smoke.txt 2>&1; tail -1 $OUT/${tag}_final_smoke.txt
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/${tag}_final_pytest_gpu.log 2>&1; tail -5 $OUT/${tag}_final_pytest_gpu.log
python bench.py --steps 5 --warmup 1 > $OUT/${tag}_final_bench.json 2> $OUT/${tag}_final_bench.err; tail -c 300 $OUT/${tag}_final_bench.err
[ -n "$FINAL_SHORT" ] && exit 0          # (smoke + suite + bench only)
export BENCH_SIZE="--rays 640000 --iters 2" BENCH_ARGS="--precision f16x3"
bash scripts/pmc_pass.sh ${tag}f_fetch FETCH_SIZE
bash scripts/pmc_pass.sh ${tag}f_write WRITE_SIZE
python - "$tag" <<'PY' > $OUT/${tag}_final_traffic.txt
import csv, glob, sys
tag = sys.argv[1]
pts = 640000 * 192
tot = {}
for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = glob.glob(f"gpurun_out/prof/{tag}f_{d}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_encode_mlp_f16x3_t128" in r["Kernel_Name"] and r["Counter_Name"] == c]
    tot[c] = sum(v) / len(v) * 1024 / 1e6
alg = pts * 48.0 + 640000 * 44 + 2.7e6
print(f"k_encode_mlp_f16x3_t128<false, false, false>, bench frame's fine launch ({pts} points), PMC in separate passes: FETCH_SIZE {tot['FETCH_SIZE']:.1f} MB + "
      f"WRITE_SIZE {tot['WRITE_SIZE']:.1f} MB = {sum(tot.values()):.1f} MB = {sum(tot.values()) * 1e6 / pts:.1f} B per point = {sum(tot.values()) * 1e6 / alg:.3f} x algorithmic ({alg / 1e6:.1f} MB)")
PY
cat $OUT/${tag}_final_traffic.txt
rm -rf $OUT/prof
( python scripts/bench_train_step.py --iters 20; python scripts/bench_train_step.py --iters 20 --ssr 28; python scripts/bench_train_step.py --iters 20 --ssr 101; python scripts/bench_ssr_frame.py --frames 6 ) 2>&1 | grep -v amdgpu | grep "training step\|SSR frame" | cut -c1-140 > $OUT/${tag}_final_steps.txt
cat $OUT/${tag}_final_steps.txt
