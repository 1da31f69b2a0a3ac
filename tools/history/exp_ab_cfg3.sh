#!/bin/bash
for L in before hip; do
  export M2S_LIB=$PWD/mesh_to_sdf_amd/libm2s_$L.so
  echo "=== lib $L"
  python tools/exp_ab_queries.py 2>&1 | grep -v amdgpu
done
