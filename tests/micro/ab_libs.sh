#!/bin/bash
# same-box A/B of library variants var_libs/lib_rx_*.so on the full bench (strip, 9000x4000 pair, chain, throughput mode): ab_libs.sh [reps]
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for rep in $(seq ${1:-3}); do for f in var_libs/lib_rx_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$f rep $rep:', r['value'], r['ms_per_step_median'], r.get('canvas_pair_9000x4000',{}).get('ms_per_pair'), r.get('config4_chain',{}).get('seconds'), r.get('throughput_mode',{}).get('value'))"
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
