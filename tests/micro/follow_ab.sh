#!/bin/bash
# flow-following window: parity (stage + batch + e2e tests), same-box A/B against the round-4 sweep: sweeps alone, lone pair, batches at x1 / x4 / x8
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=24
timeout 1500 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "throughput or latency" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_gpu_throughput.py -x -q -m gpu -k "equals_single and (8-8-2 or 6-3-2 or 12-12)" 2>&1 | tail -3
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
for v in old new; do
cp var_libs/lib_ab_$v.so panorama-opticalflow_amd/libpanoflow.so
echo "== $v"
SW_WIDE=2 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4000x128 4950x2000 2>&1 | grep "W="
SW_WIDE=0 timeout 300 python tests/micro/gpu_sweep_bench.py 4000x8 4000x32 4950x2000 2>&1 | grep "W="
timeout 900 python tests/micro/disp_probe.py 1 4 8 2>&1 | grep "in flight\|lone" | cut -c1-70
TP_PAIRS=32 TP_LOOPS=3 timeout 300 python tests/micro/throughput_one.py 32 9000 4000 2>&1 | grep queues
done 2>&1 | tee gpurun_out/r05_follow_ab.txt
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
