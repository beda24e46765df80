#!/bin/bash
# per-kernel totals of the default bench workload under rocprofv3 (GPU box): kstats.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; shift
rm -rf gpurun_out/ks_$T
env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$T -o b -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 1 > gpurun_out/ks_$T.log 2>&1
python - <<PY
import csv, glob, json
f = glob.glob('gpurun_out/ks_$T/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'pf::' in r['Name']]
line = [l for l in open('gpurun_out/ks_$T.log') if l.startswith('{"metric"')][-1]
r = json.loads(line)
print("$T $*: value %.1f Mpix/s, %.3f ms/step (under rocprof)" % (r["value"], r["ms_per_step"]))
pairs = 7.0
for x in rows[:14]:
    n = x['Name']; n = n[n.index('pf::') + 4:].split('(')[0]
    print("   %-44s calls/pair %7.1f  ms/pair %8.3f  avg_us %9.2f" % (n[:44], int(x['Calls']) / pairs, float(x['TotalDurationNs']) / 1e6 / pairs, float(x['AverageNs']) / 1e3))
PY
