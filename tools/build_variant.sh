#!/bin/bash
# Build a variant of the library with extra compile-time knobs for A/B runs (LGM_LIB=<path> selects it):
#   tools/build_variant.sh <tag> <file.hip> [-DKNOB=value ...]   ->  lightning-generative-models_amd/csrc/liblgm_hip_<tag>.so
# Only <file.hip> is recompiled with the knobs; every other object is the in-tree build's (run the normal build first).
set -e
tag=$1; src=$2; shift 2
cs=lightning-generative-models_amd/csrc
obj=/tmp/lgm_variant_${tag}_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value "$@" -c $cs/$src -o $obj
objs=$(ls $cs/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $cs/liblgm_hip_$tag.so $objs $obj
echo $cs/liblgm_hip_$tag.so
