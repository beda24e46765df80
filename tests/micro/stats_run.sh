#!/bin/bash
# per-band counters of the sweep kernel (needs var_libs/lib_stats.so = a -DPF_SWEEP_STATS -DPF_SWEEP_STATS_PRINT build): stats_run.sh WxH
cd $GRAFT_REPO_ROOT
cp panorama-opticalflow_amd/libpanoflow.so /tmp/lib_orig.so
cp var_libs/lib_stats.so panorama-opticalflow_amd/libpanoflow.so
python - <<PY
import sys, os, numpy as np
sys.path.insert(0, "tests")
from conftest import load_pkg_module
pf = load_pkg_module("pyabi")
ctx = pf.Context(0, sweep_wide=int(os.environ.get("SW_WIDE", "0")))   # (build var_libs/lib_stats.so with -DPF_EXPERIMENTS for form 1)
r = np.random.default_rng(0)
w, h = [int(v) for v in "$1".split("x")]
g0 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1
g1 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1
flow = r.standard_normal((h, w, 2)).astype(np.float32)
bl = flow * 0.9
a = np.ones((h, w), np.float32)
ctx.stage_sweep(g0, g1, bl, a, a, flow, 1)
PY
cp /tmp/lib_orig.so panorama-opticalflow_amd/libpanoflow.so
