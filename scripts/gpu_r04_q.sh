#!/bin/bash
# round 4, run q: the training part of scripts/collect_profiles.sh again on the final code (thirteen / fourteen products in the batched launch)
tag=r04
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
python scripts/bench_train_step.py --iters 8 > $OUT/${tag}_train_step.txt 2>&1
python scripts/bench_train_step.py --iters 8 --ssr 28 >> $OUT/${tag}_train_step.txt 2>&1
rm -rf $OUT/prof/${tag}_train $OUT/prof/${tag}_tr_w $OUT/prof/${tag}_tr_f $OUT/prof/${tag}_tr_m
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/${tag}_train -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
find $OUT/prof/${tag}_train -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_train_step_kernel_stats.csv \;
export BENCH_SCRIPT=scripts/bench_train_kernels.py BENCH_SIZE="--rays 2048 --iters 1" BENCH_ARGS=""
bash scripts/pmc_pass.sh ${tag}_tr_w WRITE_SIZE
bash scripts/pmc_pass.sh ${tag}_tr_f FETCH_SIZE
bash scripts/pmc_pass.sh ${tag}_tr_m GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU
python scripts/pmc_report.py $OUT/prof/${tag}_tr_w $OUT/prof/${tag}_tr_f $OUT/prof/${tag}_tr_m > $OUT/${tag}_train_pmc_summary.txt 2>&1
for p in w f m; do find $OUT/prof/${tag}_tr_$p -name "*counter_collection.csv" -exec cp {} $OUT/${tag}_train_pmc_$p.csv \; ; done
python scripts/bench_train_kernels.py > $OUT/${tag}_train_kernels.txt 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${tag}q_bench_nocpu.json 2> $OUT/${tag}q_bench_nocpu.err
grep -v amdgpu.ids $OUT/${tag}_train_step.txt; grep -v amdgpu.ids $OUT/${tag}_train_kernels.txt
python -c "
import json; d = json.load(open('$OUT/${tag}q_bench_nocpu.json')); print(d['value'], d['roofline']['frac'], d['train_step']['ms_per_step'], d['train_step']['graphed_ms_per_step'])"
python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/r04z_smoke.txt 2>&1; tail -1 $OUT/r04z_smoke.txt
timeout 1200 python -m pytest tests -m gpu -q > $OUT/r04z_pytest_gpu.log 2>&1; tail -2 $OUT/r04z_pytest_gpu.log
