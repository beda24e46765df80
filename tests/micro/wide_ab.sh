#!/bin/bash
# same-box A/B of the sweep workgroup shapes in throughput mode: 8 dense 9000x4000 pairs and 8 / 16 strips, one batch of 8
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=3
for rep in 1 2; do
for w in 0 -1 1; do
  echo "== sweep_wide=$w"
  TP_PAIRS=8 TP_WIDE=$w python tests/micro/throughput_one.py 8 9000 4000 2>&1 | grep queues
  TP_PAIRS=16 TP_WIDE=$w python tests/micro/throughput_one.py 8 2000 4000 2>&1 | grep queues
done
done
