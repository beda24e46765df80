"""durations of the last 12 kernels of stream queue 2 (level 0) from a kernel trace"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'pf::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); n = r['Kernel_Name']; n = n[n.index('pf::') + 4:]; r['k'] = n.split('(')[0].split('<')[0]
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if r['k'] == 'k_downscale_gray'][::2]
sol = rows[starts[3]:starts[4]]
qs = {}
for r in sol: qs.setdefault(r['Queue_Id'], []).append(r)
q = max(qs.values(), key=len)
t0 = q[0]['s']
print("direction span %.3f ms, n %d" % ((q[-1]['e'] - t0) / 1e6, len(q)))
for r in q[-11:]: print("   %-22s %8.1f us (gap before %5.1f)  grid %s" % (r['k'], (r['e'] - r['s']) / 1e3, 0, r['Grid_Size_X']))
tot = {}
for r in q: tot[r['k']] = tot.get(r['k'], 0) + r['e'] - r['s']
print({k: round(v / 1e6, 3) for k, v in tot.items()})
