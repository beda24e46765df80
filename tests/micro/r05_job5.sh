#!/bin/bash
# flow-following window, latency form: parity of the stage / e2e tests, displacement timing + out-of-window counts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=24
timeout 1500 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "latency" 2>&1 | tail -12 > gpurun_out/r05_job5_tests.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -12 >> gpurun_out/r05_job5_tests.txt
cat gpurun_out/r05_job5_tests.txt
DISP_INFLIGHT=0 timeout 900 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_disp_timing_follow.txt
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_stats.so panorama-opticalflow_amd/libpanoflow.so
DISP_STATS=1 timeout 600 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_disp_stats_follow.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
