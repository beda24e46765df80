#!/bin/bash
# Same-box A/B of library variants (box-to-box variation is ~2-3 %, so every kept change is measured against its
# predecessor inside ONE gpurun call).  Variants: var_libs/lib_<name>.so built beforehand in the build container, e.g.
#   git stash / checkout <ref> -- panorama-opticalflow_amd/csrc; make; cp libpanoflow.so var_libs/lib_<name>.so
# Usage on the GPU box: bash tests/micro/ab2.sh [reps]   -> per variant: value / ms per step (median of reps), 9000x4000 pair, chain
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
REPS=${1:-3}
for rep in $(seq $REPS); do for f in var_libs/lib_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  python bench.py --no-cpu-baseline --steps 10 --warmup 2 ${AB_ARGS:---no-extras} 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$f rep $rep:', r['value'], r['ms_per_step_median'], r.get('canvas_pair_9000x4000',{}).get('ms_per_pair'), r.get('config4_chain',{}).get('seconds'))"
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
