#!/bin/bash
# usage: run_variants.sh  -- tries every var_libs/lib_*.so in place of libpanoflow.so (experiment helper, not a test)
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for f in var_libs/lib_*.so; do
  echo "=== $f"
  cp $f panorama-opticalflow_amd/libpanoflow.so
  (timeout 300 python -m pytest tests/test_gpu_stages.py -q -m gpu -x -k "sweep or level" 2>&1 | tail -1)
  timeout 200 python tests/micro/gpu_sweep_bench.py 4000x8 4000x32 1100x2000 550x1000 140x250 2>&1 | tail -3
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
