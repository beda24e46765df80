#!/bin/bash
# GPU box: effective shader clock (GRBM_GUI_ACTIVE / kernel wall time, MI355X_MICROARCH.md DVFS section) of the sweep kernels --
# a lone band, and the dense 9000x4000 pair -- and what rocm-smi reports while a pair loop runs.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/clk
cat > /tmp/clk.py <<'PY'
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
sys.path.insert(0, "tests")
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
c = pf.Context(0, 9000, 4000)
rng = np.random.default_rng(5); h, w = 8, 4096
g0 = rng.standard_normal((h, w, 2)).astype(np.float32) * 0.05; g1 = rng.standard_normal((h, w, 2)).astype(np.float32) * 0.05
bl = rng.standard_normal((h, w, 2)).astype(np.float32); fl = rng.standard_normal((h, w, 2)).astype(np.float32); a = np.ones((h, w), np.float32)
for _ in range(4): c.stage_sweep(g0, g1, bl, a, a, fl, True)
dev = torch.device("cuda", 0)
L, R, b, _ = synth.make_pair(9000, 4000, 1234, dev); o = torch.empty((4000, 9000, 4), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n): c.novel_view_dev(L.data_ptr(), R.data_ptr(), 9000, 4000, 0, b.data_ptr(), o.data_ptr())
PY
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/clk -o c -- python /tmp/clk.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/clk/**/*counter_collection.csv', recursive=True)
t = glob.glob('gpurun_out/clk/**/*kernel_trace.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(t[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'], int(r['Grid_Size_X']))
rows = []
for r in csv.DictReader(open(f[0])):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d = dur.get(r['Dispatch_Id'])
    if d and 'k_sweep2' in d[1] and d[0] > 20000:
        rows.append((d[2], d[0], float(r['Counter_Value'])))
print("k_sweep2 dispatches > 20 us: grid threads, duration us, GRBM_GUI_ACTIVE cycles, effective clock GHz")
for g, ns, cyc in rows[:4] + sorted(rows, key=lambda x: -x[1])[:6]:
    print("  %8d  %9.1f  %12.0f  %.3f" % (g, ns / 1e3, cyc, cyc / ns))
PY
( python /tmp/clk.py 60 > /dev/null 2>&1 & ) ; sleep 12; rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | head -6; rocm-smi --showperflevel 2>/dev/null | grep -i perf; sleep 6
