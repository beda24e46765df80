#!/bin/bash
# parity of relaxation-sweep build variants (var_libs/lib_rx_*.so), stage tests only
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for f in var_libs/lib_rx_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  echo "== $f"; RX_QUICK=1 timeout 300 python tests/micro/relax_check.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
