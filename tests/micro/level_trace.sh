#!/bin/bash
# per-level sweep kernel durations of one flow direction on the 2000x4000 strip (rocprofv3 kernel trace; experiment helper)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/lt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt -o t -- python tests/micro/gpu_dir_probe.py > gpurun_out/lt.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('gpurun_out/lt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep2' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for call in range(3):   # the three pf_flow calls: 36 levels x (fwd, bwd) each, coarse -> fine
    first = rows[72 * call:72 * call + 72]
    tot = 0
    for i in range(0, 72, 2):
        d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in first[i:i + 2]]
        g = int(first[i]['Grid_Size_X']) // int(first[i]['Workgroup_Size_X'])
        tot += sum(d)
        if i >= 62: print("call %d level %2d wgs %3d fwd %8.1f us bwd %8.1f us" % (call, 35 - i // 2, g, d[0], d[1]))
    print("call %d total %.2f ms" % (call, tot / 1e3))
PY
