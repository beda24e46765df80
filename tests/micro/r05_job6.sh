#!/bin/bash
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
timeout 1500 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "latency" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -4
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for rep in 1 2; do for v in old new; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo "== $v rep $rep"
SW_WIDE=0 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x8 4000x32 4950x2000 2>&1 | grep "W="
DISP_INFLIGHT=0 timeout 300 python tests/micro/disp_probe.py 1 8 2>&1 | grep "lone pair" | cut -c1-60
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
