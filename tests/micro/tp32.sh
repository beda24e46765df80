#!/bin/bash
# how far does more in flight go?  dense pairs and strips, 16 / 24 / 32 in flight (batches of up to 16)
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=2
for n in 16 24 32; do
  echo -n "dense $n in flight: "; TP_PAIRS=$n python tests/micro/throughput_one.py $n 9000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
  echo -n "strips $n in flight: "; TP_PAIRS=$n python tests/micro/throughput_one.py $n 2000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
done
