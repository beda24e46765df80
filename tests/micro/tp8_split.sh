#!/bin/bash
# GPU box: 8 dense 9000x4000 pairs in flight, split into lanes of batch_pairs pairs each (and the same for 16 strips)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=${Q:-16} TP_LOOPS=3
for b in -1 8 4 2; do
  TP_PAIRS=8 TP_BATCH=$b python tests/micro/throughput_one.py 8 9000 4000 2>&1 | grep queues
done
for b in -1 16 8 4; do
  TP_PAIRS=16 TP_BATCH=$b python tests/micro/throughput_one.py 16 2000 4000 2>&1 | grep queues
done
