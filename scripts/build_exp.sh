#!/bin/bash
# builds scripts/exp/lib_<tag>.so from the csrc sources with extra flags for ds_photo.hip:  build_exp.sh <tag> <flags...>
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p scripts/exp
C=depth_estimation_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Iinclude -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c -o scripts/exp/photo_$tag.o $C/ds_photo.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o scripts/exp/lib_$tag.so scripts/exp/photo_$tag.o $C/ds_warp.o $C/ds_misc.o $C/ds_step.o
rm scripts/exp/photo_$tag.o
