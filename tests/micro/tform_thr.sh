#!/bin/bash
# throughput form: which launches should take it?  dense 9000x4000 pairs, 8 and 16 in flight, sweep_wide_threshold sweep (GPU box)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=2
for spec in "8 8" "16 16"; do
  set -- $spec
  for thr in lat 1024 512 256; do
    if [ $thr = lat ]; then W=0; T=768; else W=-1; T=$thr; fi
    echo -n "pairs $1 in_flight $2 threshold $thr: "
    TP_PAIRS=$1 TP_BATCH=8 TP_WIDE=$W TP_WIDE_THR=$T python tests/micro/throughput_one.py $2 9000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
  done
done
