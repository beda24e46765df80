cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o clk -- python tests/micro/gpu_sweep_bench.py 4000x8 1100x2000 > gpurun_out/pmc1.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/pmc1/**/*counter_collection.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    print(f, len(rows), list(rows[0].keys()) if rows else None)
    for r in rows:
        if 'sweep2' in r.get('Kernel_Name',''):
            print(r.get('Kernel_Name')[:30], r.get('Counter_Name'), r.get('Counter_Value'), int(r.get('End_Timestamp',0))-int(r.get('Start_Timestamp',0)))
PY
# sample clocks while sweeps run back-to-back
(python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, 'tests')
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); ctx = pf.Context(0)
r = np.random.default_rng(0); w,h=1100,2000
g0 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1; g1 = g0.copy(); flow = r.standard_normal((h, w, 2)).astype(np.float32); a = np.ones((h, w), np.float32)
import time; t=time.time()
while time.time()-t < 6: ctx.stage_sweep(g0, g1, flow, a, a, flow, 1)
PY
) &
sleep 3
rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk"
sleep 1
rocm-smi --showclocks 2>&1 | grep -E "sclk"
wait
