#!/bin/bash
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_tftiming.so panorama-opticalflow_amd/libpanoflow.so
SW_WIDE=2 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4950x2000 2>&1 | tr '\r' '\n' | grep "W=\|t-form" | sort | uniq -c | sort -rn | head -12
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
