#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_<name>.so (default: old = the round-4 sweep, new = this tree): follow_ab.sh [names...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=24
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for v in ${@:-old new}; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo "== $v"
SW_WIDE=2 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4950x2000 2>&1 | grep "W="
SW_WIDE=0 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x8 4000x32 4950x2000 2>&1 | grep "W="
DISP_INFLIGHT=${AB_INFLIGHT:-16} timeout 900 python tests/micro/disp_probe.py ${AB_SCALES:-1 8} 2>&1 | grep "in flight\|lone" | cut -c1-70
done 2>&1 | tee gpurun_out/follow_ab.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
