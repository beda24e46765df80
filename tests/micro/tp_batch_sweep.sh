#!/bin/bash
# GPU box: throughput mode over (in_flight, batch_pairs) splits, 24 strips of 2000x4000 (or $1 x $2)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=24
for spec in "1 1" "2 1" "2 2" "4 1" "4 2" "4 4" "6 1" "6 2" "6 3" "6 6" "8 2" "8 4" "8 8" "12 4" "12 6" "16 8"; do
  set -- $spec
  TP_BATCH=$2 timeout 120 python tests/micro/throughput_one.py $1 ${COLS:-2000} ${ROWS:-4000} 2>&1 | grep -v amdgpu.ids | tail -1
done
