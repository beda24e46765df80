#!/bin/bash
# the longest launches of every kernel family in one strip solve (level 0) + their grid sizes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fm
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fm -o t -- python bench.py --no-cpu-baseline --no-extras --no-profile --steps 3 --warmup 1 > gpurun_out/fm.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/fm/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pf::' in r['Kernel_Name']]
fam = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']; n = n[n.index('pf::') + 4:].split('(')[0]
    fam[n].append(((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], r.get('LDS_Block_Size', ''), r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('SGPR_Count', '')))
for n, v in sorted(fam.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    v.sort(reverse=True)
    print("%-34s n %5d total %8.2f ms  top3 %s  grid %sx%s wg %s lds %s vgpr %s" % (n[:34], len(v), sum(x[0] for x in v) / 1e3, " ".join("%.1f" % x[0] for x in v[:3]), v[0][1], v[0][2], v[0][3], v[0][4], v[0][5]))
PY
rm -rf gpurun_out/fm
