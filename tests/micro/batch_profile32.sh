#!/bin/bash
# kernel stats of the SATURATED throughput mode: 32 dense 9000x4000 pairs in flight (two lanes x one batch of 16)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16 TP_PAIRS=32 TP_LOOPS=1
D=gpurun_out/bp32; rm -rf $D; mkdir -p $D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o b -- python tests/micro/throughput_one.py 32 9000 4000 > $D/ks.log 2>&1
grep queues $D/ks.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/bp32/ks/**/*kernel_stats.csv', recursive=True)[0]
fam = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if 'pf::' not in r['Name']: continue
    n = r['Name']; n = n[n.index('pf::') + 4:].split('(')[0]
    n = n.split('<')[0] + ('<' + n.split('<')[1][:12] if '<' in n and ('sweep' in n) else '')
    fam[n] += float(r['TotalDurationNs'])
calls, pairs = 2, 32
print("# 32 dense 9000x4000 pairs in flight (2 lanes x 16): kernel ms per PAIR, summed over the four direction streams")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print("%-40s %8.3f" % (k, v / 1e6 / calls / pairs))
print("%-40s %8.3f" % ("sum", sum(fam.values()) / 1e6 / calls / pairs))
PY
