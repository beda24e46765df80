#!/bin/bash
# throughput form, records through LDS (3 bands per workgroup, sweep_wide 3) vs requested ahead by the compute wave (4 bands, sweep_wide 2)
cd $GRAFT_REPO_ROOT
for w in 3 2; do echo "== form $w"; SW_WIDE=$w python tests/micro/gpu_sweep_bench.py 4000x32 4000x128 4000x960 4950x2000 2>&1 | grep W=; done
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=2
for rep in 1 2; do for spec in "8 8" "16 16"; do set -- $spec
  echo -n "pairs $1 in_flight $2 auto: "; TP_PAIRS=$1 TP_BATCH=8 python tests/micro/throughput_one.py $2 9000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
done; done
