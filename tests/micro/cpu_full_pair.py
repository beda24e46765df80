"""GPU-box HOST: the oracle (CPU port of the reference's path) on the WHOLE dense 9000x4000 pair of the bench (seed 1234, pixflow_low):
2 threads (one per flow direction) then 1 thread, + the blend; writes gpurun_out/cpu_full_pair.json (round-4 review, next #6).
Timer boundaries as the reference's (CPU/main.cpp:49,62,103-108: images in memory -> composite in memory, no file I/O)."""
import hashlib, json, os, platform, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from conftest import load_pkg_module
synth = load_pkg_module("synth")
orc.build()
cols, rows = 9000, 4000
L, R, blend, _ = synth.make_pair(cols, rows, 1234, "cpu")
L, R, blend = L.numpy(), R.numpy(), blend.numpy()
res = [None, None]
def run(d): res[d] = orc.flow_one_dir(L, R, 0, d)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(d,)) for d in (0, 1)]; [t.start() for t in th]; [t.join() for t in th]
t_two = time.perf_counter() - t0
t0 = time.perf_counter(); out = orc.combine_novel_views(L, R, res[0], res[1], blend); t_blend = time.perf_counter() - t0
sha = [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (res[0], res[1], out)]
fx = np.load(os.path.join(ROOT, "tests", "golden", "dense_9000x4000.npz"))
same = sha == [str(v) for v in fx["sha_outputs"]]
one = None
if os.environ.get("CPU_FULL_ONE_THREAD", "1") == "1":
    t0 = time.perf_counter(); [orc.flow_one_dir(L, R, 0, d) for d in (0, 1)]; one = time.perf_counter() - t0
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
mp = cols * rows / 1e6
r = {"workload": "the bench's dense 9000x4000 pair (seed 1234), pixflow_low, whole path: 2 flow directions + novel-view blend", "kind": "port (oracle/pixflow_oracle.cpp, g++ -O2 -ffp-contract=off)",
     "two_threads": {"seconds": round(t_two + t_blend, 2), "flow_seconds": round(t_two, 2), "blend_seconds": round(t_blend, 2), "Mpix/s": round(mp / (t_two + t_blend), 4), "cores": 2},
     "one_thread": None if one is None else {"seconds": round(one + t_blend, 2), "Mpix/s": round(mp / (one + t_blend), 4), "cores": 1},
     "outputs_equal_committed_fixture_sha256": same, "cpu_model": cpu, "nproc": os.cpu_count(), "machine": platform.machine()}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(r, open(os.path.join(ROOT, "gpurun_out", "cpu_full_pair.json"), "w"), indent=1)
print(json.dumps(r))
