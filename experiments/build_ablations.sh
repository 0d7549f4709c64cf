#!/bin/bash
# Builds intrinsicnerf_amd/libinerf_abl<N>.so for each PIPE_ABL value given (timing experiments on the pipe kernel; see
# csrc/mlp_f16_pipe.hip).  Run after `python -m intrinsicnerf_amd._build`; load with INERF_LIB_OVERRIDE=<path>.
set -e
cd "$(dirname "$0")/../intrinsicnerf_amd"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 \
      -DPIPE_ABL=$n -c csrc/mlp_f16_pipe.hip -o csrc/_obj/pipe_abl$n.o &
done
wait
for n in "$@"; do
  objs=$(ls csrc/_obj/*.o | grep -v "mlp_f16_pipe.hip.o" | grep -v "pipe_abl")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs csrc/_obj/pipe_abl$n.o -o libinerf_abl$n.so
done
ls -la libinerf_abl*.so
