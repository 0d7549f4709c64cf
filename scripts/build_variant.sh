#!/bin/bash
# Builds intrinsicnerf_amd/libinerf_<name>.so with one source recompiled under extra flags (kernel experiments):
#   scripts/build_variant.sh <name> <source in csrc/> "<extra hipcc flags>"
# Run after `python -m intrinsicnerf_amd._build`; load with INERF_LIB_OVERRIDE=<path>.
set -e
name=$1; src=$2; extra=$3
cd "$(dirname "$0")/../intrinsicnerf_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-comment -Wno-unused-result $extra -c csrc/$src -o csrc/_obj/variant_$name.o
objs=$(ls csrc/_obj/*.o | grep -v "/$src.o" | grep -v "variant_")
# the binding resolves inerf_build_digest(); a variant carries its name instead of a digest (only INERF_LIB_OVERRIDE loads it)
echo "extern \"C\" const char* inerf_build_digest(void) { return \"variant:$name\"; }" > csrc/_obj/variant_digest_$name.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -x c++ csrc/_obj/variant_digest_$name.cpp -x none $objs csrc/_obj/variant_$name.o -o libinerf_$name.so
rm -f csrc/_obj/variant_digest_$name.cpp
ls -la libinerf_$name.so
