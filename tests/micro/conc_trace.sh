#!/bin/bash
# kernel execution times (rocprofv3 kernel trace) with 1 and 2 pairs in flight: is the slowdown of the wide kernels execution or queueing?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in 1 2; do
  rm -rf gpurun_out/ct$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ct$n -o t -- python bench.py --no-cpu-baseline --no-profile --steps 3 --concurrent $n > gpurun_out/ct$n.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob('gpurun_out/ct$n/**/*kernel_stats.csv', recursive=True)[0]
print("concurrent $n")
for r in csv.DictReader(open(f)):
    n_ = r['Name']
    if 'pf::' in n_ and any(k in n_ for k in ('k_median5', 'k_gauss15_row', 'k_upsample', 'k_sweep2<true, true', 'k_sweep_prep')):
        print("  %-40s calls %6s avg_us %9.2f" % (n_.split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
