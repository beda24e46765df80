"""Summarises batch_profile.sh's rocprofv3 output: per kernel family, per PAIR of the 8-pair batch: launches, ms, algorithmic
bytes (SURVEY.md 8(d) model), GB/s, share of the batch's wall time; plus the PMC traffic per family."""
import collections, csv, glob, json, os, re, sys

D, R, cols, rows = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
NP, CALLS = 8, 3   # pairs per batch call; throughput_one.py makes 1 warm-up + TP_LOOPS=2 timed calls
pad = cols // 20; ce = cols + 2 * pad
w, h = int(ce * 0.5), int(rows * 0.5)
P = 0
while w > 24 and h > 24:
    P += w * h; w, h = int(w * 0.9 + 0.5), int(h * 0.9 + 0.5)
# algorithmic bytes per pair and family (both directions): SURVEY 8(d)
ALG = {"k_sweep": 2 * 96 * P, "k_median5": 2 * 32 * P, "k_gauss15": 2 * 64 * P, "k_gradients": 24 * P, "k_pyr": 35.75 * P,
       "k_upsample": 2 * 14.5 * P, "k_final_flow": 2 * 28.6 * cols * rows, "k_blend": 32 * cols * rows, "k_downscale": 8.8 * cols * rows,
       "k_gauss_small": 4.4 * cols * rows, "k_gate": 9 * P}
def fam_of(name):
    n = name[name.index("pf::") + 4:] if "pf::" in name else name
    n = n.split("(")[0].split("<")[0]
    for k in ALG:
        if n.startswith(k): return k
    return n
ks = glob.glob(os.path.join(D, "ks", "**", "*kernel_stats.csv"), recursive=True)[0]
fam = collections.defaultdict(lambda: {"calls": 0, "ns": 0.0})
for r in csv.DictReader(open(ks)):
    if "pf::" not in r["Name"]: continue
    f = fam_of(r["Name"]); fam[f]["calls"] += int(r["Calls"]); fam[f]["ns"] += float(r["TotalDurationNs"])
line = [l for l in open(os.path.join(D, "ks.log")) if l.startswith("queues")][-1].strip()
plain = [l for l in open(os.path.join(D, "plain.log")) if l.startswith("queues")][-1].strip()
pmc = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
have_pmc = False
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(D, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pf::" not in r["Kernel_Name"]: continue
            pmc[fam_of(r["Kernel_Name"])][c] += float(r["Counter_Value"]) * 1024; have_pmc = True
out = ["# %s: ONE batch of %d pairs %dx%d through pf_novel_view_batch_dev (in_flight 8 = one batch), rocprofv3 --kernel-trace --stats" % (R, NP, cols, rows),
       "# unprofiled: " + plain, "# under rocprofv3: " + line,
       "# per PAIR (totals / %d calls / %d pairs); kernel ms are SUMMED over concurrent streams (two directions overlap), GB/s = algorithmic bytes / kernel ms" % (CALLS, NP),
       "%-22s %10s %10s %12s %10s %14s %14s" % ("family", "launches", "ms/pair", "alg MB/pair", "alg GB/s", "fetch MB raw", "write MB")]
tot = 0.0
js = {}
for f, v in sorted(fam.items(), key=lambda kv: -kv[1]["ns"]):
    ms = v["ns"] / 1e6 / CALLS / NP; tot += ms
    alg = ALG.get(f)
    fe = pmc[f]["FETCH_SIZE"] / NP / 1e6 if have_pmc else None; wr = pmc[f]["WRITE_SIZE"] / NP / 1e6 if have_pmc else None   # PMC passes run ONE call
    out.append("%-22s %10.1f %10.3f %12s %10s %14s %14s" % (f, v["calls"] / CALLS, ms, "%.0f" % (alg / 1e6) if alg else "-",
               "%.0f" % (alg / ms / 1e6) if alg else "-", "%.0f" % fe if fe is not None else "-", "%.0f" % wr if wr is not None else "-"))
    js[f] = {"launches_per_call": v["calls"] / CALLS, "ms_per_pair": ms, "alg_bytes_per_pair": alg, "fetch_bytes_raw_per_pair": fe and fe * 1e6, "write_bytes_per_pair": wr and wr * 1e6}
out.append("%-22s %10s %10.3f" % ("sum of kernel time", "", tot))
txt = "\n".join(out); print(txt)
open("gpurun_out/%s_batch_summary_%d.txt" % (R, cols), "w").write(txt + "\n")
import csv as _csv   # (without the rows of torch's own kernels, which make the synthetic inputs)
_rows = list(_csv.reader(open(ks)))
_csv.writer(open("gpurun_out/%s_batch_kernel_stats_%d.csv" % (R, cols), "w", newline="")).writerows(
    [_rows[0]] + [r for r in _rows[1:] if not any(t in r[0] for t in ("at::", "rocprim", "rocclr", "hipcub", "elementwise", "c10::"))])
if have_pmc:
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) of ONE pf_novel_view_batch_dev call on 8 pairs %dx%d; bytes per PAIR; gfx950 counts wide reads at 1/2: true fetch in [raw, 2*raw]" % (cols, rows),
               "families": js}, open("gpurun_out/%s_batch_pmc_%d.json" % (R, cols), "w"), indent=1)
