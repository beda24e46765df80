#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_*.so: lone pair / strip (ab_time.py) and 8 dense pairs + 16 strips in flight: ab_thr.sh [reps]
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
export GPU_MAX_HW_QUEUES=16 TP_LOOPS=3
for rep in $(seq ${1:-1}); do for f in var_libs/lib_ab_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  echo "== $(basename $f) rep $rep"
  timeout 300 python tests/micro/ab_time.py "$(basename $f)" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
  TP_PAIRS=8 timeout 300 python tests/micro/throughput_one.py 8 9000 4000 2>&1 | grep queues
  TP_PAIRS=16 timeout 300 python tests/micro/throughput_one.py 16 2000 4000 2>&1 | grep queues
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
