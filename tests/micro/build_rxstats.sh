#!/bin/bash
# build var_libs/lib_rxstats.so (the relaxation sweep with -DPF_RX_STATS); run from the repo root after `make`
set -e
cd "$(dirname "$0")/../../panorama-opticalflow_amd"
mkdir -p ../var_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -DPF_RX_STATS $RX_DEFS -c csrc/kernels_sweep2.hip -o /tmp/sw2_stats.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../var_libs/lib_rxstats.so csrc/pf_api.o csrc/pf_dist.o csrc/kernels_pre.o csrc/kernels_level.o csrc/kernels_sweep.o /tmp/sw2_stats.o csrc/kernels_misc.o -ldl
