"""Per-stream timeline of one steady-state solve from a rocprofv3 --kernel-trace CSV (tests/micro/kstats.sh output):
when each stream's first kernel starts relative to the solve's first kernel, busy time, gaps, and the first kernels."""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'pf::' in r['Kernel_Name']]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); n = r['Kernel_Name']; n = n[n.index('pf::') + 4:]; r['k'] = n.split('(')[0].split('<')[0]
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if r['k'] == 'k_downscale_gray'][::2]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[which], starts[which + 1]
sol = rows[a:b]
t0 = sol[0]['s']; tend = max(r['e'] for r in sol)
print("solve span %.3f ms, %d kernels" % ((tend - t0) / 1e6, len(sol)))
byq = collections.defaultdict(list)
for r in sol:
    byq[r['Queue_Id']].append(r)
for q, lst in sorted(byq.items(), key=lambda kv: kv[1][0]['s']):
    busy = sum(r['e'] - r['s'] for r in lst); gaps = [lst[i + 1]['s'] - lst[i]['e'] for i in range(len(lst) - 1)]
    print("queue %s: n %d first %.3f ms last end %.3f ms busy %.3f gaps %.3f (median %.2f us)" % (q, len(lst), (lst[0]['s'] - t0) / 1e6, (lst[-1]['e'] - t0) / 1e6, busy / 1e6, sum(gaps) / 1e6, (sorted(gaps)[len(gaps) // 2] / 1e3) if gaps else 0))
    for r in lst[:int(sys.argv[3]) if len(sys.argv) > 3 else 6]:
        print("      %-22s start %8.1f us dur %7.1f us" % (r['k'], (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3))
    r = lst[-1]
    print("      ... last: %-16s start %8.1f us dur %7.1f us" % (r['k'], (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3))
