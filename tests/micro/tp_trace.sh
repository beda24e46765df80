#!/bin/bash
# throughput mode under a kernel trace: are the kernels slower with more pairs in flight, or the gaps between them longer?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
for k in 1 2 4; do
  rm -rf gpurun_out/tp_$k
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tp_$k -o t -- python tests/micro/throughput_one.py $k > gpurun_out/tp_$k.log 2>&1
  grep queues gpurun_out/tp_$k.log
  python - <<PY
import csv, glob, collections
f = glob.glob('gpurun_out/tp_$k/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pf::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); n = r['Kernel_Name']; n = n[n.index('pf::') + 4:]; r['k'] = n.split('(')[0].split('<')[0]
rows.sort(key=lambda r: r['s'])
rows = rows[len(rows) * 2 // 3:]          # the last of the three batch calls
span = (rows[-1]['e'] - rows[0]['s']) / 1e6
tot = collections.defaultdict(float); cnt = collections.Counter(); mx = collections.defaultdict(float)
for r in rows: d = (r['e'] - r['s']) / 1e3; tot[r['k']] += d; cnt[r['k']] += 1; mx[r['k']] = max(mx[r['k']], d)
pairs = cnt['k_blend'] if 'k_blend' in cnt else 24
print("  in_flight $k: span %.1f ms, %d pairs; per pair: " % (span, pairs) + ", ".join("%s %.2f ms (max %.0f us)" % (k_, tot[k_] / 1e3 / pairs, mx[k_]) for k_ in sorted(tot, key=lambda x: -tot[x])[:6]))
qs = collections.defaultdict(list)
for r in rows: qs[r['Queue_Id']].append(r)
busy = {q: sum(r['e'] - r['s'] for r in v) / 1e6 for q, v in qs.items()}
print("  queues: " + ", ".join("%s: %d kernels busy %.1f ms" % (q, len(qs[q]), busy[q]) for q in sorted(qs)))
PY
done
