#!/bin/bash
# throughput mode with batches under a kernel trace: per hardware queue, how much of the span is covered by its kernels (union of the
# kernel intervals: kernels of one stream do not overlap each other), for round 2's split (lanes x 1) and round 3's (one batch).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32
for spec in "1 1" "6 1" "6 6" "8 8"; do
  set -- $spec
  rm -rf gpurun_out/tpb_$1_$2
  TP_BATCH=$2 timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tpb_$1_$2 -o t -- python tests/micro/throughput_one.py $1 > gpurun_out/tpb_$1_$2.log 2>&1
  grep queues gpurun_out/tpb_$1_$2.log
  python - "$1" "$2" <<'PY'
import csv, glob, collections, sys
k, b = sys.argv[1], sys.argv[2]
f = glob.glob('gpurun_out/tpb_%s_%s/**/*kernel_trace.csv' % (k, b), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pf::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
rows = rows[len(rows) * 2 // 3:]          # the last of the three batch calls
t0, t1 = rows[0]['s'], rows[-1]['e']
span = (t1 - t0) / 1e6
qs = collections.defaultdict(list)
for r in rows: qs[r['Queue_Id']].append((r['s'], r['e']))
out = []
for q, iv in sorted(qs.items()):
    iv.sort(); cov = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: cov += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    cov += ce - cs
    if len(iv) > 50: out.append("q%s %d kernels busy %.0f %%" % (q, len(iv), 100.0 * cov / (t1 - t0)))
print("  in_flight %s, %s pairs per batch: span %.1f ms for 24 strips; launches %d; %s" % (k, b, span, len(rows), "; ".join(out)))
PY
done
