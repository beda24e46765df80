cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcb_$c
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcb_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile > gpurun_out/pmcb_$c.log 2>&1
  echo "$c rc=$?"; tail -2 gpurun_out/pmcb_$c.log | cut -c1-300
done
python - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob('gpurun_out/pmcb_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:44]
            if k.startswith('void at::') or 'rocclr' in k: continue
            tot[k] += float(r['Counter_Value']); n[k] += 1
    print(c, "sum KB over all dispatches")
    for k in sorted(tot, key=lambda k: -tot[k])[:12]:
        print("  %-46s dispatches %5d  sum_KB %14.0f" % (k, n[k], tot[k]))
PY
