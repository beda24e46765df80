#!/bin/bash
# GPU box: per-launch durations of one kernel family grouped by grid size (rocprofv3 kernel trace of two dense 9000x4000 pairs):
#   kern_by_grid.sh <kernel-name-substring> [top-n]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
K=${1:-k_median5}
rm -rf gpurun_out/kbg
cat > /tmp/kbg.py <<'PY'
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, "tests")
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0)
cc, cr = 9000, 4000
c = pf.Context(0, cc, cr)
L, R, b, _ = synth.make_pair(cc, cr, 1234, dev); o = torch.empty((cr, cc, 4), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
for _ in range(2): c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, b.data_ptr(), o.data_ptr())
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kbg -o t -- python /tmp/kbg.py > /dev/null 2>&1
python - "$K" "${2:-12}" <<'PY'
import csv, glob, sys, collections
k, topn = sys.argv[1], int(sys.argv[2])
f = glob.glob('gpurun_out/kbg/**/*kernel_trace.csv', recursive=True)[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if k in r['Kernel_Name']:
        g[(int(r['Grid_Size_X']), int(r['Grid_Size_Y']), int(r['Workgroup_Size_X']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in g.values())
print("%s: %d launches, %.3f ms total" % (k, sum(len(v) for v in g.values()), tot / 1e3))
for key, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:topn]:
    print("  grid %s x %s (wg %d): n %d  avg %.1f us  min %.1f" % (key[0], key[1], key[2], len(v), sum(v) / len(v), min(v)))
PY
