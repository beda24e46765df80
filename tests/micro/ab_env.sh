#!/bin/bash
# same-box A/B of an environment switch: ab_env.sh VAR VAL_A VAL_B [reps] [bench args...]
cd $GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; REPS=${4:-3}; shift 4
for rep in $(seq $REPS); do for val in $A $B; do
  env $V=$val python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels_ms_per_step']; print('$V=$val rep $rep:', r['value'], r['ms_per_step_median'], 'sweep', k.get('sweep'))"
done; done
