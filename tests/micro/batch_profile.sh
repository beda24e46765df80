#!/bin/bash
# Per-kernel evidence for the THROUGHPUT MODE (GPU box): rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes, of pf_novel_view_batch_dev on 8 same-size pairs solved as ONE batch (every launch carries the 8 pairs).
#   batch_profile.sh <tag> [cols rows [pmc]]   ->  gpurun_out/<tag>_batch_kernel_stats.csv, <tag>_batch_summary.txt, <tag>_batch_pmc.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r04}; COLS=${2:-9000}; ROWS=${3:-4000}; PMC=${4:-1}
export GPU_MAX_HW_QUEUES=${BP_QUEUES:-32} TP_PAIRS=8 TP_LOOPS=2
D=gpurun_out/bp_${R}_${COLS}
rm -rf $D; mkdir -p $D
python tests/micro/throughput_one.py 8 $COLS $ROWS > $D/plain.log 2>&1; grep queues $D/plain.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o b -- python tests/micro/throughput_one.py 8 $COLS $ROWS > $D/ks.log 2>&1
grep queues $D/ks.log
if [ "$PMC" = "1" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    TP_LOOPS=0 timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D/pmc_$c -o p -- python tests/micro/throughput_one.py 8 $COLS $ROWS > $D/pmc_$c.log 2>&1
    echo "$c rc=$?"
  done
fi
python tests/micro/batch_summary.py $D $R $COLS $ROWS
rm -rf $D
