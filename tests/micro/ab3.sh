#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_*.so with tests/micro/ab_time.py: ab3.sh [reps]
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for rep in $(seq ${1:-2}); do for f in var_libs/lib_ab_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  timeout 300 python tests/micro/ab_time.py "$(basename $f) rep $rep" 2>&1 | grep -v amdgpu.ids | tail -1
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
