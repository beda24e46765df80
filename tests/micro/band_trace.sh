#!/bin/bash
# per-band progress traces of the latency-form sweep (var_libs/lib_trace.so = kernels_sweep2.hip with -DPF_SWEEP_STATS -DPF_SWEEP_TRACE): band_trace.sh WxH ...
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/${TRACE_LIB:-lib_trace}.so panorama-opticalflow_amd/libpanoflow.so
for s in "$@"; do python tests/micro/band_trace.py $s 2>&1 | grep -v "^\(RCCL\|HIP\|ROCm\|Hostname\|Librccl\)"; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
