#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
mkdir -p gpurun_out
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_stats.so panorama-opticalflow_amd/libpanoflow.so
DISP_STATS=1 timeout 900 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_disp_stats_follow.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
