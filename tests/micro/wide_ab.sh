#!/bin/bash
# same-box A/B of the sweep forms in throughput mode (one batch of 8 pairs): sweep_wide = 0 latency / -1 auto (throughput form for
# oversubscribed launches) / 2 throughput form everywhere / 1 latency step with two compute waves per SIMD
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=3
for rep in 1 2; do
for w in ${WIDE_SET:-0 -1 2}; do
  echo "== sweep_wide=$w"
  TP_PAIRS=8 TP_WIDE=$w python tests/micro/throughput_one.py 8 9000 4000 2>&1 | grep queues
  TP_PAIRS=16 TP_WIDE=$w python tests/micro/throughput_one.py 8 2000 4000 2>&1 | grep queues
done
done
