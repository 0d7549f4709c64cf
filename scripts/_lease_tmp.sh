export TMPDIR=/tmp; REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r05q -o t -- python $REPO/scripts/bench_ssr_frame.py --frames 6 --classes 28 > /dev/null 2>&1 )
find $OUT/prof/r05q -name "*kernel_stats.csv" -exec cp {} $OUT/r05q_ssr_frame_kernel_stats.csv \;
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/r05q2 -o t -- python $REPO/scripts/bench_ssr_frame.py --frames 6 --classes 101 > /dev/null 2>&1 )
find $OUT/prof/r05q2 -name "*kernel_stats.csv" -exec cp {} $OUT/r05q_ssr101_frame_kernel_stats.csv \;
rm -rf $OUT/prof
