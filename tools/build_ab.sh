#!/bin/bash
# A/B builds of the library with a macro changed in ONE translation unit: tools/build_ab.sh <tag> <unit.hip> "<-D flags>"
# -> mesh_to_sdf_amd/ab/libm2s_<tag>.so (git-ignored, travels with gpurun); use with M2S_LIB=... (tools/exp_ab.py)
set -e
TAG=$1; UNIT=$2; FLAGS=$3
cd $(dirname $0)/../mesh_to_sdf_amd/csrc
mkdir -p ../ab
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
/opt/rocm/bin/hipcc $BASE $FLAGS -c $UNIT -o ../ab/${UNIT%.hip}_$TAG.o
OBJS=""
for o in bvh sign distance sortlib sortlib_query serde client gltf tuning capi capi_io multi; do
  if [ "$o.hip" == "$UNIT" ]; then OBJS="$OBJS ../ab/${o}_$TAG.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libm2s_$TAG.so $OBJS -ldl
echo built mesh_to_sdf_amd/ab/libm2s_$TAG.so
