#!/bin/bash
# round-5 GPU job 2: VALU issue-rate micro (per-SIMD spans), displacement timing on an idle host, new tests (warning, config5_strong)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 tests/micro/valu_issue > gpurun_out/r05_valu_issue.txt 2>&1
export GPU_MAX_HW_QUEUES=24
timeout 900 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_disp_timing.txt
unset GPU_MAX_HW_QUEUES
timeout 1500 python -m pytest tests/test_gpu_hygiene.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_job2_tests.txt
cat gpurun_out/r05_valu_issue.txt | cut -c1-200
cat gpurun_out/r05_disp_timing.txt gpurun_out/r05_job2_tests.txt
