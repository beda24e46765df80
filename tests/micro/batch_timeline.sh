#!/bin/bash
# GPU box: kernel timeline of ONE 8-pair batch (9000x4000): per stream, busy time by kernel family and what overlaps what.
#   batch_timeline.sh [in_flight] [batch_pairs]   -> gpurun_out/batch_timeline_pf.csv (the pf:: rows of the rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
INFL=${1:-8}; export TP_BATCH=${2:--1}
export GPU_MAX_HW_QUEUES=8 TP_PAIRS=$INFL TP_LOOPS=1
D=gpurun_out/btl; rm -rf $D; mkdir -p $D
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $D -o b -- python tests/micro/throughput_one.py $INFL 9000 4000 > $D/run.log 2>&1
grep queues $D/run.log
f=$(find $D -name '*kernel_trace.csv' | head -1)
head -1 $f > gpurun_out/batch_timeline_pf.csv
grep 'pf::' $f >> gpurun_out/batch_timeline_pf.csv
rm -rf $D
wc -l gpurun_out/batch_timeline_pf.csv
