#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
timeout 1500 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "large_smooth" 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "large_displacement" 2>&1 | tail -12
