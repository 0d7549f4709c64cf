export BENCH_SCRIPT=scripts/bench_layered.py BENCH_SIZE="--only 256 256 --iters 3" BENCH_ARGS=""
timeout 300 bash scripts/pmc_pass.sh lay_A GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
timeout 300 bash scripts/pmc_pass.sh lay_B SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES
python scripts/pmc_report.py gpurun_out/prof/lay_A gpurun_out/prof/lay_B 2>&1 | grep -i "linear\|prof/"
tail -3 gpurun_out/prof/lay_A.log
