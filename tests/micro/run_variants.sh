cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_base.so
for v in base noreset noprefetch nouse nofeed noslow; do
  if [ $v != base ]; then cp tests/micro/lib_$v.so panorama-opticalflow_amd/libpanoflow.so; else cp /tmp/lib_base.so panorama-opticalflow_amd/libpanoflow.so; fi
  echo "== $v"; timeout 120 python tests/gpu_sweep_bench.py 4000x32 2>&1 | tail -1
done
cp /tmp/lib_base.so panorama-opticalflow_amd/libpanoflow.so
