#!/bin/bash
# timeline of the shared front end of one steady-state strip solve (first 70 kernels of the main stream + when the direction streams start)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ft
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ft -o t -- python bench.py --no-cpu-baseline --no-extras --no-profile --steps 4 --warmup 1 > gpurun_out/ft.log 2>&1
python tests/micro/timeline.py $(find gpurun_out/ft -name '*kernel_trace.csv' | head -1) 3 ${1:-70}
rm -rf gpurun_out/ft
