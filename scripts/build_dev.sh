#!/bin/bash
# Dev aid: build/lib_<tag>.so = the in-tree library with ONE translation unit recompiled with extra flags.
#   bash scripts/build_dev.sh <tag> <file.hip> [-DFLAG ...]
set -eu
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p build
python -m iffnerf_amd.build > /dev/null
obj=build/${src%.hip}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DNDEBUG "$@" -c iffnerf_amd/csrc/$src -o $obj
others=$(ls iffnerf_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_$tag.so $others $obj
echo build/lib_$tag.so
