#!/bin/bash
# Round 6, lease 5: the SSR network on the 128-point tile - parity tests, then same-box A/B of the SSR frame and of the kernel
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_size_properties.py tests/test_unfiltered_parity.py tests/test_wrappers_gpu.py tests/test_coalesce_gpu.py tests/test_range_fallback_gpu.py tests/test_train_masks_gpu.py -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15 ) > $OUT/r06_l5_tests.txt 2>&1; tail -8 $OUT/r06_l5_tests.txt
for rep in 1 2 3; do
  for f in t128 csplit; do
    echo "[$f $rep frame] $(INERF_F16_KERNEL=$f python scripts/bench_ssr_frame.py --frames 6 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170)"
    echo "[$f $rep kernel] $(INERF_F16_KERNEL=$f python scripts/bench_mlp.py --iters 5 --ssr 28 --rays 131072 --precision f16x3 2>&1 | tail -1 | cut -c1-200)"
  done
done > $OUT/r06_ssr_t128_ab.txt 2>&1
cat $OUT/r06_ssr_t128_ab.txt
for c in 5 28 32; do echo "[C=$c default] $(python scripts/bench_ssr_frame.py --frames 4 --classes $c 2>&1 | grep -v amdgpu | tail -1 | cut -c1-170)"; done >> $OUT/r06_ssr_t128_ab.txt 2>&1
tail -3 $OUT/r06_ssr_t128_ab.txt
INERF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-baseline-quick > $OUT/r06_bench_n2_shared.json 2> $OUT/r06_bench_n2_shared.err; tail -c 600 $OUT/r06_bench_n2_shared.err; python -c "
import json; d=json.loads(open('$OUT/r06_bench_n2_shared.json').read().strip().split(chr(10))[-1]); print('n2 shared: value', d['value'], 'n_gpus', d['n_gpus'], 'ref chunking', d['reference_chunking']['bit_identical_to_the_timed_frame'], 'ssr', d['configs']['ssr_room0_320x240'].get('checksum_identical_on_all_ranks'))"
