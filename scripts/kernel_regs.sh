#!/bin/bash
# Compile ONE kernel source for gfx950 the way _build.py does and print every kernel's register / scratch / LDS notes
# (the numbers tests/test_isa_audit_cpu.py asserts on).  usage: scripts/kernel_regs.sh mlp_f16.hip [extra hipcc flags]
set -e
src=$1; shift
dir=$(dirname "$0")/../intrinsicnerf_amd/csrc
extra=""
[ "$src" = "mlp_bwd.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
out=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -Wno-unused-result $extra "$@" \
    -c "$dir/$src" -o "$out/o.o" -save-temps=obj 2>&1 | grep -v "^$" | head -40
asm=$(ls "$out"/*gfx950*.s 2>/dev/null | head -1); [ -z "$asm" ] && { echo "compile failed"; rm -rf "$out"; exit 1; }
awk '/^\s*\.amdhsa_kernel /{k=$2} /\.sgpr_spill_count|\.vgpr_spill_count|\.vgpr_count|\.private_segment_fixed_size|\.name:/{print}' "$asm" \
  | paste - - - - - | sed 's/  */ /g' | c++filt | cut -c1-260
rm -rf "$out"
