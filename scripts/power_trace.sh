#!/bin/bash
# Samples socket power and clocks (rocm-smi) twice a second while the encode+MLP kernel runs back to back, per kernel form:
#   scripts/power_trace.sh [forms...]          (GPU box; writes gpurun_out/power_trace.txt)
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/power_trace.txt
: > $OUT
rocm-smi --showmaxpower >> $OUT 2>&1
for form in ${@:-dual pipe}; do
  echo "=== form $form" >> $OUT
  INERF_F16_KERNEL=$form python $REPO/scripts/bench_mlp.py --rays 640000 --iters 24 > $REPO/gpurun_out/power_trace_$form.log 2>&1 &
  pid=$!
  sleep 6          # import + warm-up
  for i in $(seq 1 14); do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level" | tr '\n' ';' >> $OUT
    echo >> $OUT
    sleep 0.5
  done
  wait $pid
  tail -1 $REPO/gpurun_out/power_trace_$form.log >> $OUT
done
echo "=== idle" >> $OUT
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $OUT
cat $OUT
