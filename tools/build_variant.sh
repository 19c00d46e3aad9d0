#!/bin/bash
# build_variant.sh <name> <unit> "<extra hipcc flags>": an A/B build of the library -- the translation unit <unit> (e.g. c2_api_count) compiled
# with the extra flags, linked with the other units' objects of the regular build -> crispresso2_amd/lib/variants/lib_<name>.so
# (run it on the GPU box with C2_AMD_LIB=<that path>; *.so files travel with gpurun and stay out of git)
set -e
NAME=$1; UNIT=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/crispresso2_amd/csrc"
make -j8 >/dev/null
mkdir -p ../lib/variants ../lib/obj_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include $FLAGS -c $UNIT.hip -o ../lib/obj_variants/${UNIT}_$NAME.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../lib/obj_variants/${UNIT}_$NAME.o -lz -lpthread -o ../lib/variants/lib_$NAME.so
echo "../lib/variants/lib_$NAME.so"
