#!/bin/bash
# same-box A/B with the per-family breakdown (HIP events) for a given size: ab_fam.sh COLS ROWS
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for f in var_libs/lib_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --cols $1 --rows $2 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$f', r['value'], r['ms_per_step_median'], json.dumps(r['kernels_ms_per_step']))"
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
