#!/bin/bash
# per-level span of one direction stream of the 2000x4000 strip: sweeps vs other kernels vs gaps (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ls
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ls -o t -- python bench.py --no-cpu-baseline --no-extras --no-profile --steps 3 --warmup 1 > gpurun_out/ls.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/ls/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pf::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); n = r['Kernel_Name']; n = n[n.index('pf::') + 4:]; r['k'] = n.split('(')[0]
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if r['k'].startswith('k_downscale_gray')][::2]
call = rows[starts[-2]:starts[-1]]          # one complete timed pair
qs = collections.defaultdict(list)
for r in call: qs[r['Queue_Id']].append(r)
q = sorted(qs.values(), key=len)[-1]        # one direction stream
lv = [i for i, r in enumerate(q) if r['k'].startswith('k_gauss15_fused<false>')] + [len(q)]
print("level  span_us  sweeps_us  other_us  gaps_us  kernels")
tot = collections.Counter()
for j in range(len(lv) - 1):
    ks = q[lv[j]:lv[j + 1]]
    end = q[lv[j + 1]]['s'] if lv[j + 1] < len(q) else ks[-1]['e']
    span = (end - ks[0]['s']) / 1e3
    sw = sum(r['e'] - r['s'] for r in ks if 'k_sweep2' in r['k']) / 1e3
    ot = sum(r['e'] - r['s'] for r in ks if 'k_sweep2' not in r['k']) / 1e3
    level = len(lv) - 2 - j
    tot['span'] += span; tot['sw'] += sw; tot['ot'] += ot
    print("%5d %8.1f %10.1f %9.1f %8.1f %5d" % (level, span, sw, ot, span - sw - ot, len(ks)))
print("total span %.2f ms  sweeps %.2f  other %.2f  gaps %.2f" % (tot['span'] / 1e3, tot['sw'] / 1e3, tot['ot'] / 1e3, (tot['span'] - tot['sw'] - tot['ot']) / 1e3))
PY
rm -rf gpurun_out/ls
