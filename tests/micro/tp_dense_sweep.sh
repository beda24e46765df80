#!/bin/bash
# throughput mode on DENSE 9000x4000 pairs: pairs in flight x batch split (GPU box)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=2
for spec in "8 8 8" "16 16 8" "16 12 6" "24 24 8" "16 16 4"; do
  set -- $spec
  TP_PAIRS=$1 TP_BATCH=$3 python tests/micro/throughput_one.py $2 9000 4000 2>&1 | grep queues
done
