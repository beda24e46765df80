#!/bin/bash
# per-tile counters of the relaxation sweep (needs var_libs/lib_rxstats.so = a -DPF_RX_STATS build): one 2000x4000 bidirectional solve
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_rxstats.so panorama-opticalflow_amd/libpanoflow.so
PANOFLOW_SWEEP=3 python - <<PY > gpurun_out/rx_stats_raw.txt 2>&1
import sys, os, numpy as np
sys.path.insert(0, "tests")
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
ctx = pf.Context(0)
L, R, blend = synth.make_pair_np(2000, 4000, 1234)
ctx.flow(L, R, 0, 0)
PY
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
grep -c "^RX" gpurun_out/rx_stats_raw.txt
