#!/bin/bash
# same-box A/B of library variants (var_libs/lib_rx_*.so) in throughput mode: 1, 2 and 6 strips in flight, twice
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for rep in 1 2; do for f in var_libs/lib_rx_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  for k in 1 2 6; do timeout 300 python tests/micro/throughput_one.py $k 2>&1 | grep queues | sed "s#^#$f rep $rep: #"; done
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
