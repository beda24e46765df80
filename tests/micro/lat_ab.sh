#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_<name>.so on the lone-pair path: latency-form sweeps + lone dense pair + strip: lat_ab.sh names...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=24
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for v in "$@"; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo -n "== $v: "
SW_WIDE=0 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4950x2000 2>&1 | grep "W=" | sed 's/.*= *\([0-9.]*\) us.*/\1/' | tr '\n' ' '
DISP_INFLIGHT=0 timeout 600 python tests/micro/disp_probe.py 1 2>&1 | grep "lone" | sed 's/.*lone pair: \([0-9.]*\) ms.*/pair \1 ms/'
done 2>&1 | tee gpurun_out/lat_ab.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
