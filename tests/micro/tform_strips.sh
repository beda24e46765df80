#!/bin/bash
# throughput form on 2000x4000 strips (transposed sweeps): 16 strips, 8 in flight (one batch) and 16 in flight (two lanes)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=2 TP_PAIRS=16
for infl in 8 16; do
  for spec in "0 0 512" "-1 1 512" "-1 1 256" "-1 1 128"; do
    set -- $spec
    echo -n "in_flight $infl sweep_wide $1 transposed $2 threshold $3: "
    TP_BATCH=8 TP_WIDE=$1 TP_WIDE_TR=$2 TP_WIDE_THR=$3 python tests/micro/throughput_one.py $infl 2000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
  done
done
