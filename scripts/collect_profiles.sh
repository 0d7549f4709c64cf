#!/bin/bash
# Collects the rocprofv3 evidence a round commits under profiles/ (run on the GPU box through gpurun; ~20 minutes with the GPU suite):
#   scripts/collect_profiles.sh r02
# Writes gpurun_out/<tag>_*; copy what is to be kept into profiles/.
tag=${1:-rXX}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
mkdir -p $OUT/prof
cd $REPO
python -c 'import __graft_entry__ as g; g.build()' || exit 1
# 0. the GPU suite on this code
timeout 1200 python -m pytest tests -m gpu -q -x > $OUT/${tag}_pytest_gpu.log 2>&1; tail -3 $OUT/${tag}_pytest_gpu.log
# 1. the benchmark as the driver runs it (with the CPU oracle: parity + cpu_baseline)
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${tag}_bench.json 2> $OUT/${tag}_bench.err
# 2. the same under rocprofv3 --kernel-trace --stats (no CPU legs: the profiler only sees the GPU)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/${tag}_bench -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${tag}_bench_under_rocprof.json 2> $OUT/${tag}_bench_under_rocprof.err )
find $OUT/prof/${tag}_bench -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_bench_kernel_stats.csv \;
# 3. PMC passes on the dominant kernel at the benchmark's launch shape (640 000 rays x 192 samples), one counter group per pass
export BENCH_SIZE="--rays 640000 --iters 2" BENCH_ARGS="--precision f16x3"
bash scripts/pmc_pass.sh ${tag}_dual_A GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
bash scripts/pmc_pass.sh ${tag}_dual_fetch FETCH_SIZE
bash scripts/pmc_pass.sh ${tag}_dual_write WRITE_SIZE
bash scripts/pmc_pass.sh ${tag}_dual_cache TCC_HIT_sum TCC_MISS_sum
python scripts/pmc_report.py $OUT/prof/${tag}_dual_A $OUT/prof/${tag}_dual_fetch $OUT/prof/${tag}_dual_write $OUT/prof/${tag}_dual_cache > $OUT/${tag}_mlp_pmc_summary.txt 2>&1
for p in A fetch write cache; do find $OUT/prof/${tag}_dual_$p -name "*counter_collection.csv" -exec cp {} $OUT/${tag}_mlp_pmc_dual_$p.csv \; ; done
# 4. the other kernel forms on the same box (same launch shape)
for f in t128 dual single; do echo "$f: $(INERF_F16_KERNEL=$f python scripts/bench_mlp.py --rays 640000 --iters 3 2>&1 | tail -1)"; done > $OUT/${tag}_kernel_forms.txt
# 5. SSR frame and the training step
python scripts/bench_ssr_frame.py --frames 5 > $OUT/${tag}_ssr_frame.txt 2>&1
python scripts/bench_train_step.py --iters 8 > $OUT/${tag}_train_step.txt 2>&1
python scripts/bench_train_step.py --iters 8 --ssr 28 >> $OUT/${tag}_train_step.txt 2>&1
# 6. the training step under rocprofv3, and PMC passes on the training kernels (fine-pass batch: 2048 rays x 192 samples)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/${tag}_train -o t -- python $REPO/scripts/bench_train_step.py --iters 8 > /dev/null 2>&1 )
find $OUT/prof/${tag}_train -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_train_step_kernel_stats.csv \;
export BENCH_SCRIPT=scripts/bench_train_kernels.py BENCH_SIZE="--rays 2048 --iters 1" BENCH_ARGS=""
bash scripts/pmc_pass.sh ${tag}_tr_w WRITE_SIZE
bash scripts/pmc_pass.sh ${tag}_tr_f FETCH_SIZE
bash scripts/pmc_pass.sh ${tag}_tr_m GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU
python scripts/pmc_report.py $OUT/prof/${tag}_tr_w $OUT/prof/${tag}_tr_f $OUT/prof/${tag}_tr_m > $OUT/${tag}_train_pmc_summary.txt 2>&1
for p in w f m; do find $OUT/prof/${tag}_tr_$p -name "*counter_collection.csv" -exec cp {} $OUT/${tag}_train_pmc_$p.csv \; ; done
python scripts/bench_train_kernels.py > $OUT/${tag}_train_kernels.txt 2>&1
# 6b. the fp32 layer kernels (csrc/layered.hip): single launches and whole networks, and one counter pass on a 256 x 256 layer
python scripts/bench_layered.py > $OUT/${tag}_layered.txt 2>&1
export BENCH_SCRIPT=scripts/bench_layered.py BENCH_SIZE="--only 256 256 --iters 3" BENCH_ARGS=""
timeout 300 bash scripts/pmc_pass.sh ${tag}_lay GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python scripts/pmc_report.py $OUT/prof/${tag}_lay k_linear > $OUT/${tag}_layered_pmc.txt 2>&1
# 7. two ranks sharing this GPU over gloo (sharding + gather logic of bench.py --gpus N, both configs; numbers mean nothing): plain
#    `python bench.py --gpus 2` - it starts its own ranks through torch.distributed.run on 127.0.0.1
INERF_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-baseline-quick > $OUT/${tag}_bench_n2_shared.json 2> $OUT/${tag}_bench_n2_shared.err
# 8. trained-network evidence
timeout 900 python scripts/fit_synthetic.py --out $OUT/${tag}_trained_network.txt > $OUT/${tag}_fit.log 2>&1
# 9. full-frame PSNR delta (needs gpurun_in/psnr_full_frame_oracle.npz from `scripts/psnr_full_frame.py --oracle` in the build container)
if [ -f gpurun_in/psnr_full_frame_oracle.npz ]; then timeout 600 python scripts/psnr_full_frame.py --hip > $OUT/${tag}_psnr_full_frame.log 2>&1; fi
tail -c 600 $OUT/${tag}_bench.err; tail -3 $OUT/${tag}_kernel_forms.txt; tail -2 $OUT/${tag}_ssr_frame.txt; tail -2 $OUT/${tag}_train_step.txt
