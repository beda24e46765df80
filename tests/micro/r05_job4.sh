#!/bin/bash
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for lib in lib_tftiming; do
cp var_libs/$lib.so panorama-opticalflow_amd/libpanoflow.so
for f in 2 4; do echo "== $lib form $f"; SW_WIDE=$f timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 2>&1 | tr '\r' '\n' | grep "W=\|loader\|compute" | tail -3; done
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
