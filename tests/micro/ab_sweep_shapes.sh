#!/bin/bash
# same-box comparison of library variants var_libs/lib_ab_*.so on tests/micro/gpu_sweep_bench.py shapes: ab_sweep_shapes.sh [WxH ...]
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for f in var_libs/lib_ab_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  echo "== $(basename $f)"
  timeout 300 python tests/micro/gpu_sweep_bench.py "$@" 2>&1 | grep "us/step"
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
