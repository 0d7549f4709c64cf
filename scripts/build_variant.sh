#!/bin/bash
# Builds intrinsicnerf_amd/libinerf_<name>.so with one or more sources recompiled under extra flags (kernel experiments):
#   scripts/build_variant.sh <name> <source in csrc/>[,<source>...] "<extra hipcc flags>"
# Run after `python -m intrinsicnerf_amd._build`; load with INERF_LIB_OVERRIDE=<path>.
set -e
name=$1; srcs=$2; extra=$3
cd "$(dirname "$0")/../intrinsicnerf_amd"
objs=$(ls csrc/_obj/*.o | grep -v "variant_")
vobjs=""
for src in ${srcs//,/ }; do
  fl="$extra"; [ "$src" = "mlp_bwd.hip" ] && fl="$extra -mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -Wno-unused-result $fl -c csrc/$src -o csrc/_obj/variant_${name}_$src.o &
  objs=$(echo "$objs" | grep -v "/$src.o")
  vobjs="$vobjs csrc/_obj/variant_${name}_$src.o"
done
wait
# the binding resolves inerf_build_digest(); a variant carries its name instead of a digest (only INERF_LIB_OVERRIDE loads it)
echo "extern \"C\" const char* inerf_build_digest(void) { return \"variant:$name\"; }" > csrc/_obj/variant_digest_$name.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -x c++ csrc/_obj/variant_digest_$name.cpp -x none $objs $vobjs -o libinerf_$name.so
rm -f csrc/_obj/variant_digest_$name.cpp
ls -la libinerf_$name.so
