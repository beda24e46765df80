cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for rep in 1; do for f in var_libs/lib_*.so; do
  echo "=== $f"
  cp $f panorama-opticalflow_amd/libpanoflow.so
  timeout 200 python tests/micro/gpu_sweep_bench.py 4000x32 1100x2000 2>&1 | tail -2
  timeout 200 python tests/micro/gpu_dir_probe.py 2>&1 | tail -2
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
