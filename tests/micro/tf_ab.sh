#!/bin/bash
# same-box A/B of library variants var_libs/lib_ab_<name>.so: throughput-form sweep alone + dense pairs 8 / 32 in flight: tf_ab.sh names...
cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=32 TP_LOOPS=3
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for v in "$@"; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo "== $v"
SW_WIDE=2 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4950x2000 2>&1 | grep "W="
for n in 8 32; do echo -n "dense $n in flight: "; TP_PAIRS=$n python tests/micro/throughput_one.py $n 9000 4000 2>&1 | grep queues | sed 's/.*in_flight/in_flight/'; done
done
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
