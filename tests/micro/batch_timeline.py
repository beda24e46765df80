"""Reads gpurun_out/batch_timeline_pf.csv (tests/micro/batch_timeline.sh): the LAST batch call's per-stream busy time by kernel family and the
wall-clock share of each combination of families in flight."""
import csv, collections, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/batch_timeline_pf.csv'
rows = list(csv.DictReader(open(f)))
def fam(n):
    k = re.search(r'pf::(\w+)', n).group(1)
    if k.startswith('k_sweep_t'): return 'sweepT'
    if k.startswith('k_sweep2'): return 'sweepL'
    if k.startswith('k_sweep_prep'): return 'prep'
    if k.startswith('k_median'): return 'median'
    if k.startswith('k_gauss15'): return 'gauss'
    return 'other'
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['f'] = fam(r['Kernel_Name']); r['k'] = re.search(r'pf::(\w+)', r['Kernel_Name']).group(1)
rows.sort(key=lambda r: r['s'])
bl = [i for i, r in enumerate(rows) if r['k'].startswith('k_blend')]
call = rows[bl[-2] + 1: bl[-1] + 1] if len(bl) > 1 else rows[:bl[-1] + 1]
t0 = min(r['s'] for r in call); t1 = max(r['e'] for r in call)
print("call span %.2f ms, %d kernels" % ((t1 - t0) / 1e6, len(call)))
bys = collections.defaultdict(list)
for r in call: bys[r['Stream_Id']].append(r)
for s, v in sorted(bys.items()):
    busy = sum(r['e'] - r['s'] for r in v)
    fc = collections.Counter()
    for r in v: fc[r['f']] += r['e'] - r['s']
    gaps = sum(max(0, v[i + 1]['s'] - v[i]['e']) for i in range(len(v) - 1))
    print("stream %s: n %d start %.2f span %.2f busy %.2f gaps %.2f | " % (s, len(v), (v[0]['s'] - t0) / 1e6, (v[-1]['e'] - v[0]['s']) / 1e6, busy / 1e6, gaps / 1e6)
          + "  ".join("%s %.2f" % (k, t / 1e6) for k, t in fc.most_common()))
# wall-clock share by set of families in flight
ev = []
for r in call: ev.append((r['s'], 1, r['f'])); ev.append((r['e'], -1, r['f']))
ev.sort()
cur = collections.Counter(); last = t0; share = collections.Counter()
for t, d, fm in ev:
    key = "+".join(sorted(k for k, c in cur.items() if c > 0)) or "idle"
    share[key] += t - last; last = t
    cur[fm] += d
for k, t in share.most_common(20): print("  %-40s %.2f ms" % (k, t / 1e6))
