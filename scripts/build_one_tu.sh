#!/bin/bash
# Dev aid: a variant library that differs from the product in ONE translation unit (the others are the in-tree objects).
#   bash scripts/build_one_tu.sh TAG TU.hip [extra hipcc flags]   ->  build/lib_TAG.so   (load it with IFF_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
tag=$1; tu=$2; shift 2
python -m iffnerf_amd.build > /dev/null
mkdir -p build/$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -DNDEBUG \
  -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c iffnerf_amd/csrc/$tu -o build/$tag/${tu%.hip}.o 2> >(grep -v "not a recognized feature" >&2)
others=$(ls iffnerf_amd/csrc/*.o | grep -v "/${tu%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_$tag.so $others build/$tag/${tu%.hip}.o
echo build/lib_$tag.so
