"""Diagnostics for DESIGN.md (not a test): host CPU, measured HBM copy bandwidth, CPU-port timings.

Run on the GPU box:  python tests/micro/gpu_host_info.py   ("confirm the peak with a device
stream benchmark", "state nproc, CPU model and both [1-thread, 2-thread] numbers").
"""
import json, os, subprocess, sys, threading, time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg_module  # noqa: E402

out = {}
try:
    lscpu = subprocess.run(["lscpu"], capture_output=True, text=True).stdout
    for line in lscpu.splitlines():
        if line.startswith(("Model name", "CPU(s):", "Thread(s) per core", "Socket(s)", "CPU max MHz")):
            k, v = line.split(":", 1)
            out[k.strip()] = v.strip()
except Exception as e:  # noqa: BLE001
    out["lscpu"] = str(e)
out["nproc"] = os.cpu_count()
out["sched_affinity"] = len(os.sched_getaffinity(0))

# ---- HBM: device-to-device copy of 2 GiB (reads + writes counted), and a read-only reduction ----
dev = torch.device("cuda:0")
n = 2 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
out["hbm_copy_GBps"] = round(2 * n / ms / 1e6, 1)
af = a.view(torch.float32)
for _ in range(2):
    af.sum()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    af.sum()
e1.record(); torch.cuda.synchronize()
out["hbm_read_GBps"] = round(n / (e0.elapsed_time(e1) / 10) / 1e6, 1)
out["device"] = torch.cuda.get_device_name(0)
del a, b, af

# ---- CPU port on the strip: 1 thread (reference has no threading) and 2 threads (one per direction) ----
if "--no-cpu" not in sys.argv:
    synth = load_pkg_module("synth")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.build()
    L, R, _ = synth.make_pair_np(2000, 4000, 1234)
    t = time.perf_counter()
    f0 = orc.flow_one_dir(L, R, 0, 0)
    t1dir = time.perf_counter() - t
    t = time.perf_counter()
    f1 = orc.flow_one_dir(L, R, 0, 1)
    t1 = t1dir + (time.perf_counter() - t)
    res = {}
    def run(d):
        res[d] = orc.flow_one_dir(L, R, 0, d)
    t = time.perf_counter()
    th = [threading.Thread(target=run, args=(d,)) for d in (0, 1)]
    [x.start() for x in th]; [x.join() for x in th]
    t2 = time.perf_counter() - t
    out["cpu_port_strip_1thread_s"] = round(t1, 2)
    out["cpu_port_strip_2thread_s"] = round(t2, 2)
    out["cpu_port_strip_1thread_Mpix_s"] = round(8.0 / t1, 4)
    out["cpu_port_strip_2thread_Mpix_s"] = round(8.0 / t2, 4)
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "host_info.json"), "w"), indent=1)
