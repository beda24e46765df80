#!/bin/bash
# round-5 GPU job 3: fused-prepass throughput form -- parity (stage + batch tests of the form) and same-box A/B in the batch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=24
timeout 1200 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "throughput_fused" 2>&1 | tail -8 > gpurun_out/r05_job3_tests.txt
timeout 1200 python -m pytest tests/test_gpu_throughput.py -x -q -m gpu -k "equals_single and (8-4 or 5-4 or 3-4 or 12-4)" 2>&1 | tail -8 >> gpurun_out/r05_job3_tests.txt
cat gpurun_out/r05_job3_tests.txt
for f in 2 3 4; do echo "== form $f"; SW_WIDE=$f timeout 300 python tests/micro/gpu_sweep_bench.py 4000x32 4000x128 4000x960 4950x2000 2>&1 | grep "W="; done | tee gpurun_out/r05_fused_sweep.txt
export TP_LOOPS=3
for rep in 1 2; do for fused in 0 1; do
  TP_PAIRS=8 TP_FUSED=$fused timeout 300 python tests/micro/throughput_one.py 8 9000 4000 2>&1 | grep queues
  TP_PAIRS=32 TP_FUSED=$fused timeout 300 python tests/micro/throughput_one.py 32 9000 4000 2>&1 | grep queues
done; done 2>&1 | tee gpurun_out/r05_fused_ab.txt
