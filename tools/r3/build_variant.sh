#!/bin/bash
# tools/r3/build_variant.sh <name> <conv_igemm source> [extra hipcc flags]: a copy of libblades_hip.so with another
# conv_igemm object -> tools/ab_lib/libblades_hip_<name>.so (same-box A/B through BLADES_HIP_LIB)
set -e
name=$1; src=$2; shift 2
mkdir -p tools/ab_lib /tmp/abv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /tmp/abv/conv_igemm_$name.o -I blades_amd/ops/csrc -Wno-unused-result "$@" 2>/dev/null
objs=$(ls blades_amd/ops/_build/*.o | grep -v conv_igemm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab_lib/libblades_hip_$name.so $objs /tmp/abv/conv_igemm_$name.o
echo built tools/ab_lib/libblades_hip_$name.so
