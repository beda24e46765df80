#!/bin/bash
# GPU box: kernel timeline of the config-4 chain (5 x pf_stitch_step, 9000x4000): chain_timeline.sh -> gpurun_out/chain_timeline_pf.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/ctl; rm -rf $D; mkdir -p $D
cat > /tmp/ctl.py <<PY
import os, sys, time, numpy as np, torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "tests")
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cc, cr = 9000, 4000
top, imgs = synth.make_stitch_set(cc, cr, 1234, 5, torch.device("cuda", 0))
top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
c = pf.Context(0, cc, cr)
final = np.zeros((cr, cc, 4), np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    for i, im in enumerate(imgs):
        c.stitch_prefetch(None if i == 4 else imgs[i + 1])
        c.stitch_step(im, top if i == 0 else None, 20, want_out=(i == 4), out=final if i == 4 else None)
    print("chain %.1f ms" % (1000 * (time.perf_counter() - t0)))
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o b -- python /tmp/ctl.py > $D/run.log 2>&1
grep chain $D/run.log
f=$(find $D -name '*kernel_trace.csv' | head -1)
head -1 $f > gpurun_out/chain_timeline_pf.csv
grep 'pf::' $f | tail -n 5000 >> gpurun_out/chain_timeline_pf.csv
rm -rf $D
wc -l gpurun_out/chain_timeline_pf.csv
