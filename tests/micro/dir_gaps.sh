#!/bin/bash
# gaps between a direction stream's kernels: one direction alone vs both directions side by side (2000x4000 strip)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/dg
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dg -o t -- python tests/micro/gpu_dir_probe.py > gpurun_out/dg.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/dg/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pf::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); n = r['Kernel_Name']; n = n[n.index('pf::') + 4:]; r['k'] = n.split('(')[0].split('<')[0]
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if r['k'] == 'k_downscale_gray'][::2] + [len(rows)]
for c in range(len(starts) - 1):
    call = rows[starts[c]:starts[c + 1]]
    qs = collections.defaultdict(list)
    for r in call: qs[r['Queue_Id']].append(r)
    out = []
    for q, v in sorted(qs.items(), key=lambda kv: -len(kv[1]))[:2]:
        if len(v) < 100: continue
        span = (v[-1]['e'] - v[0]['s']) / 1e6; busy = sum(r['e'] - r['s'] for r in v) / 1e6
        out.append("queue %s: %d kernels, span %.2f ms, busy %.2f ms, gaps %.2f ms (%.2f us per kernel)" % (q, len(v), span, busy, span - busy, 1000 * (span - busy) / len(v)))
    print("call %d: total span %.2f ms | " % (c, (call[-1]['e'] - call[0]['s']) / 1e6) + " | ".join(out))
PY
rm -rf gpurun_out/dg
