#!/bin/bash
# build var_libs/lib_rx_<name>.so with extra -D flags: build_rxvar.sh <name> [-D...]
set -e
cd "$(dirname "$0")/../../panorama-opticalflow_amd"
N=$1; shift
mkdir -p ../var_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c csrc/kernels_sweep2.hip -o /tmp/sw2_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../var_libs/lib_rx_$N.so csrc/pf_api.o csrc/pf_dist.o csrc/kernels_pre.o csrc/kernels_level.o csrc/kernels_sweep.o /tmp/sw2_$N.o csrc/kernels_misc.o -ldl
