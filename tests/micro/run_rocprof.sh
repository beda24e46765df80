# rocprofv3 evidence for the bench (run on the GPU box through gpurun); summaries are copied into profiles/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r01}
rm -rf gpurun_out/prof_$R gpurun_out/pmc_$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$R.log 2>&1
tail -1 gpurun_out/prof_$R.log | cut -c1-300
python - <<PY
import csv, glob
for f in glob.glob('gpurun_out/prof_$R/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f)
    for r in rows[:16]:
        print("%-60s calls %6s total_ms %10.3f avg_us %10.2f pct %6s" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
# HBM traffic counters in their own passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$R/fetch -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/pmc_$R.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$R/write -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/pmc_$R.write.log 2>&1
python - <<PY
import csv, glob, collections
for kind in ("fetch", "write"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob('gpurun_out/pmc_$R/%s/**/*counter_collection.csv' % kind, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:40]
            tot[k] += float(r['Counter_Value']); n[k] += 1
    print(kind, "KB per kernel family (sum over all dispatches of warmup+1 step):")
    for k in sorted(tot, key=lambda k: -tot[k])[:14]:
        print("  %-42s dispatches %6d  sum %14.0f KB" % (k, n[k], tot[k]))
PY
