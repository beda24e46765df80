#!/bin/bash
# stand-alone durations of the stencil kernels at the strip's level-0 size (nothing else on the GPU)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st -o t -- python tests/micro/stage_times.py "$@" > gpurun_out/st.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/st/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'pf::' in r['Name']: print("%-60s calls %4s avg %8.1f us  min %8.1f" % (r['Name'][r['Name'].index('pf::'):][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
rm -rf gpurun_out/st
