#!/bin/bash
# 16 dense pairs / 16 strips in flight: two lanes x 8 pairs per batch vs ONE batch of 16 (kMaxBatch 16)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=2 TP_PAIRS=16
for rep in 1 2; do for b in 8 16; do
  echo -n "dense 16 in flight, batch_pairs $b: "; TP_BATCH=$b python tests/micro/throughput_one.py 16 9000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
  echo -n "strips 16 in flight, batch_pairs $b: "; TP_BATCH=$b python tests/micro/throughput_one.py 16 2000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
done; done
