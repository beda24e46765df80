#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for f in var_libs/lib_rx_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  rm -rf gpurun_out/ag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ag -o t -- python bench.py --no-cpu-baseline --no-extras --no-profile --steps 5 --warmup 1 > gpurun_out/ag.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob('gpurun_out/ag/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_gate_bbox_all' in r['Name']: print("$f: k_gate_bbox_all avg %.1f us min %.1f" % (float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
done
rm -rf gpurun_out/ag
for rep in 1 2 3; do for f in var_libs/lib_rx_*.so; do
  cp $f panorama-opticalflow_amd/libpanoflow.so
  python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('$f rep $rep:', r['value'], r['ms_per_step_median'])"
done; done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
