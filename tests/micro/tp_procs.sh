#!/bin/bash
# throughput with lanes as separate PROCESSES (one pair in flight each) instead of threads of one process
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for n in 1 2 4; do
  echo "== $n processes x in_flight 1"
  for i in $(seq $n); do TP_LOOPS=6 timeout 300 python tests/micro/throughput_one.py 1 2>&1 | grep queues & done
  wait
done
