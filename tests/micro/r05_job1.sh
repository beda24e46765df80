#!/bin/bash
# round-5 GPU job 1: VALU issue-rate micro, displacement sensitivity (timing + out-of-window counts), full-pair CPU baseline (host cores, beside)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python tests/micro/cpu_full_pair.py > gpurun_out/cpu_full_pair.log 2>&1) &
CPUJOB=$!
timeout 300 tests/micro/valu_issue > gpurun_out/r05_valu_issue.txt 2>&1
export GPU_MAX_HW_QUEUES=24
timeout 900 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_disp_timing.txt
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_stats.so panorama-opticalflow_amd/libpanoflow.so
DISP_STATS=1 timeout 600 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_disp_stats.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
wait $CPUJOB
cat gpurun_out/r05_valu_issue.txt | head -70
cat gpurun_out/r05_disp_timing.txt gpurun_out/r05_disp_stats.txt
tail -3 gpurun_out/cpu_full_pair.log
