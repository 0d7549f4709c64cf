#!/bin/bash
# Socket power and clock (rocm-smi, 2 Hz) while scripts/microbench/mfma_peak.hip issues nothing but v_mfma_f32_32x32x16_f16:
# one / two waves per SIMD, random / zero operands, ~4-7 s each (GPU box; writes gpurun_out/power_trace_mfma.txt)
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/power_trace_mfma.txt
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/scripts/microbench/mfma_peak.hip -o mfma_peak || exit 1
: > $OUT
./mfma_peak 20000000 > $REPO/gpurun_out/power_trace_mfma.log 2>&1 &
pid=$!
t0=$(date +%s.%N)
while kill -0 $pid 2>/dev/null; do
  awk -v a=$(date +%s.%N) -v b=$t0 'BEGIN { printf "t=%5.1f s  ", a - b }' >> $OUT
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Current Socket Graphics Package Power|sclk clock level" | sed 's/GPU\[0\]\t\t: //' | tr '\n' ';' >> $OUT
  echo >> $OUT
  sleep 0.4
done
cat $REPO/gpurun_out/power_trace_mfma.log >> $OUT
cat $OUT
