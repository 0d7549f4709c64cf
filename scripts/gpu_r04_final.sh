#!/bin/bash
# round 4, last run: GPU suite, smoke() and the bench line on the final code
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build(); g.smoke(); print("smoke ok")' > $OUT/r04f_smoke.txt 2>&1; tail -2 $OUT/r04f_smoke.txt
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/r04f_pytest_gpu.log 2>&1; tail -3 $OUT/r04f_pytest_gpu.log
python bench.py --steps 5 --warmup 1 > $OUT/r04f_bench.json 2> $OUT/r04f_bench.err; tail -c 300 $OUT/r04f_bench.err
python scripts/bench_train_step.py --iters 8 > $OUT/r04f_train_step.txt 2>&1
python scripts/bench_train_step.py --iters 8 --ssr 28 >> $OUT/r04f_train_step.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r04f_train -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
find $OUT/prof/r04f_train -name "*kernel_stats.csv" -exec cp {} $OUT/r04f_train_step_kernel_stats.csv \;
python scripts/bench_train_kernels.py > $OUT/r04f_train_kernels.txt 2>&1
grep -v amdgpu.ids $OUT/r04f_train_step.txt; python -c "
import json; d = json.load(open('$OUT/r04f_bench.json')); print(d['value'], d['roofline']['frac'], d['train_step']['ms_per_step'], d['train_step']['graphed_ms_per_step'], d['parity']['stagewise_violations'])"
rm -rf $OUT/prof
