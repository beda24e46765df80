#!/bin/bash
# same-box A/B: the workgroup's last compute wave stores its hand-off granules itself (lib_ab_selfpub.so, -DPF_SELF_PUBLISH=1) against the publisher wave
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_ab_selfpub.so panorama-opticalflow_amd/libpanoflow.so
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "latency and (sweep or level)" 2>&1 | tail -3
for rep in 1 2; do for v in new selfpub; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo "== $v rep $rep"
SW_WIDE=0 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4000x960 4950x2000 2>&1 | grep "W="
DISP_INFLIGHT=0 timeout 900 python tests/micro/disp_probe.py 1 2>&1 | grep "lone" | cut -c1-70
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
