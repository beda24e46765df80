#!/bin/bash
# GPU box: kernel timeline of ONE lone pair (default 9000x4000): pair_timeline.sh [cols rows] -> gpurun_out/pair_timeline_pf.csv
# (the pf:: rows of the rocprofv3 kernel trace; tests/micro/pair_timeline.py reads it)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=${1:-9000}; R=${2:-4000}
D=gpurun_out/ptl; rm -rf $D; mkdir -p $D
cat > /tmp/ptl.py <<PY
import os, sys, time, torch
sys.path.insert(0, "tests")
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0); cc, cr = $C, $R
c = pf.Context(0, cc, cr)
L, R, b, _ = synth.make_pair(cc, cr, 1234, dev); o = torch.empty((cr, cc, 4), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
for i in range(3):
    t = time.perf_counter(); c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, b.data_ptr(), o.data_ptr()); print("pair %.3f ms" % (1000 * (time.perf_counter() - t)))
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o b -- python /tmp/ptl.py > $D/run.log 2>&1
grep "pair" $D/run.log
f=$(find $D -name '*kernel_trace.csv' | head -1)
head -1 $f > gpurun_out/pair_timeline_pf.csv
grep 'pf::' $f >> gpurun_out/pair_timeline_pf.csv
rm -rf $D
wc -l gpurun_out/pair_timeline_pf.csv
