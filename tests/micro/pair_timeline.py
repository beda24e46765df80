"""Reads gpurun_out/pair_timeline_pf.csv (tests/micro/pair_timeline.sh): the LAST lone-pair call: per stream busy time by kernel, gaps between a
stream's kernels, and the sweep launches in order (duration against steps x t_step when the level geometry is known)."""
import csv, collections, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pair_timeline_pf.csv'
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['k'] = re.search(r'pf::(\w+)', r['Kernel_Name']).group(1)
rows.sort(key=lambda r: r['s'])
bl = [i for i, r in enumerate(rows) if r['k'].startswith('k_blend')]
call = rows[bl[-2] + 1: bl[-1] + 1]
t0 = min(r['s'] for r in call); t1 = max(r['e'] for r in call)
print("call span %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(call)))
bys = collections.defaultdict(list)
for r in call: bys[r['Stream_Id']].append(r)
for s, v in sorted(bys.items()):
    busy = sum(r['e'] - r['s'] for r in v)
    fc = collections.Counter(); nc = collections.Counter()
    for r in v: fc[r['k']] += r['e'] - r['s']; nc[r['k']] += 1
    gaps = [max(0, v[i + 1]['s'] - v[i]['e']) for i in range(len(v) - 1)]
    print("stream %s: n %d start %.3f end %.3f busy %.3f gaps %.3f (n>2us: %d, max %.1f us)" % (s, len(v), (v[0]['s'] - t0) / 1e6, (v[-1]['e'] - t0) / 1e6, busy / 1e6, sum(gaps) / 1e6, sum(g > 2000 for g in gaps), max(gaps) / 1e3 if gaps else 0))
    for k, t in fc.most_common(): print("      %-28s n %4d  %.3f ms  avg %.1f us" % (k, nc[k], t / 1e6, t / nc[k] / 1e3))
if '-sweeps' in sys.argv:
    # level geometry of the dense pair: W0 = int((C + 2 * (C // 20)) / 2), H0 = R // 2, x 0.9 per level while both stay >= the reference's minimum
    for s, v in sorted(bys.items()):
        sw = [r for r in v if r['k'].startswith('k_sweep2') or r['k'].startswith('k_sweep_t')]
        if not sw: continue
        print("stream", s)
        for r in sw: print("   %-12s grid %6s x %3s  %.1f us" % (r['k'], r['Grid_Size_X'], r['Grid_Size_Y'], (r['e'] - r['s']) / 1e3))
