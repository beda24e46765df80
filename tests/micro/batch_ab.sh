#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_<name>.so in the throughput mode: dense pairs 8 / 16 / 32 in flight, 16 strips: batch_ab.sh names...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=3
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for v in "$@"; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo "== $v"
for n in ${AB_NS:-8 16 32}; do
  echo -n "dense $n in flight: "; TP_PAIRS=$n TP_CHECK=${AB_CHECK:-} python tests/micro/throughput_one.py $n 9000 4000 2>&1 | grep "queues\|DIFFER\|differ" | sed 's/.*in_flight/in_flight/'
done
echo -n "strips 16 in flight: "; TP_PAIRS=16 python tests/micro/throughput_one.py 16 2000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'
done 2>&1 | tee gpurun_out/batch_ab.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
