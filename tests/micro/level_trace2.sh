#!/bin/bash
# per-level sweep kernel durations of one flow direction on the 2000x4000 strip: level_trace2.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
rm -rf gpurun_out/lt_$T
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt_$T -o t -- python tests/micro/gpu_dir_probe.py > gpurun_out/lt_$T.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/lt_$T/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep2' in r['Kernel_Name'] or 'k_sweep_relax' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // 3
first = rows[2 * n:3 * n]
tot = 0
for i in range(0, n, 2):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in first[i:i + 2]]
    g = int(first[i]['Grid_Size_X']) // int(first[i]['Workgroup_Size_X'])
    tot += sum(d)
    print("$T level %2d wgs %3d fwd %8.1f us bwd %8.1f us" % (n // 2 - 1 - i // 2, g, d[0], d[1]))
print("$T total %.2f ms over %d launches" % (tot / 1e3, n))
PY
