export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; tag=r06f
cd $REPO
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${tag}_bench.json 2> $OUT/${tag}_bench.err; echo rc=$?
rm -rf $OUT/prof/${tag}_bench
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/${tag}_bench -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/${tag}_bench_under_rocprof.err )
find $OUT/prof/${tag}_bench -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_bench_kernel_stats.csv \;
