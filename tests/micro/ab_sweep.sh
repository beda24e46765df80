#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_*.so: per-step time of the sweep kernel for a few shapes + tests/micro/ab_time.py: ab_sweep.sh [reps]
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for rep in $(seq ${1:-1}); do for f in var_libs/lib_ab_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  echo "== $(basename $f) rep $rep"
  timeout 300 python tests/micro/gpu_sweep_bench.py 4000x8 4000x32 4000x512 4950x2000 2000x4950 602x244 2>&1 | grep -v amdgpu.ids
  timeout 300 python tests/micro/ab_time.py "$(basename $f) rep $rep" 2>&1 | grep -v amdgpu.ids | tail -1
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
