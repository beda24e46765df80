#!/bin/bash
# DYNAMIC instruction counts per kernel family of the throughput mode (8 dense pairs, one batch): rocprofv3 --pmc SQ_INSTS_* (one pass,
# --kernel-trace only), summed per family and divided by the 8 pairs -- which kernels the chip's VALU issue capacity goes to
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8 TP_PAIRS=8 TP_LOOPS=0
D=gpurun_out/insts; rm -rf $D; mkdir -p $D
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D/$tag -o p -- python tests/micro/throughput_one.py 8 9000 4000 > $D/$tag.log 2>&1
  echo "$set rc=$?"
done
python - <<'PY'
import csv, glob, collections
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/insts/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pf::' not in r['Kernel_Name']: continue
        n = r['Kernel_Name']; n = n[n.index('pf::') + 4:].split('(')[0]
        if 'k_sweep' in n: n = n.split(',')[0] + ('>' if '<' in n else '')
        else: n = n.split('<')[0]
        fam[n][r['Counter_Name']] += float(r['Counter_Value'])
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
print("# ONE pf_novel_view_batch_dev call on 8 dense 9000x4000 pairs; per PAIR, in millions (wave-instructions; *_CYCLES in quad-cycles per the guide)")
print("%-34s" % "kernel" + "".join("%14s" % c.replace("SQ_", "") for c in cols))
tot = collections.defaultdict(float)
for k, v in sorted(fam.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    print("%-34s" % k[:34] + "".join("%14.1f" % (v.get(c, 0) / 8 / 1e6) for c in cols))
    for c in cols: tot[c] += v.get(c, 0)
print("%-34s" % "total" + "".join("%14.1f" % (tot[c] / 8 / 1e6) for c in cols))
PY
rm -rf $D   # the raw counter files are tens of MB: gpurun only copies back 64 MiB
