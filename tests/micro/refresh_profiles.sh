#!/bin/bash
# Regenerates the rocprofv3 evidence behind bench.py's roofline object (run on the GPU box through gpurun):
#   gpurun_out/<R>_bench_kernel_stats.csv, <R>_bench_line.json  (--kernel-trace --stats on the default bench command)
#   gpurun_out/<R>_pmc_bench.json                               (--pmc FETCH_SIZE / WRITE_SIZE, separate passes)
# Copy them into profiles/ afterwards.  Counters are never combined with sys/hip/hsa tracing.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r01}
rm -rf gpurun_out/prof_$R gpurun_out/pmcb_*
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o bench -- python bench.py --no-cpu-baseline --no-extras --pairs-total 0 > gpurun_out/prof_$R.log 2>&1
grep "^{\"metric\"" gpurun_out/prof_$R.log | tail -1 > gpurun_out/${R}_bench_line.json
# (the synthetic inputs are made with torch on the GPU: its at::native / rocprim kernels are not the product's -- filtered out, percentages left as reported)
python - <<PY
import csv, glob
src = sorted(glob.glob('gpurun_out/prof_$R/**/*kernel_stats.csv', recursive=True))[0]
rows = list(csv.reader(open(src)))
keep = [rows[0]] + [r for r in rows[1:] if not any(t in r[0] for t in ('at::', 'rocprim', 'rocclr', 'hipcub', 'elementwise', 'c10::'))]
csv.writer(open('gpurun_out/${R}_bench_kernel_stats.csv', 'w', newline='')).writerows(keep)
print("kernel stats: kept %d of %d rows" % (len(keep) - 1, len(rows) - 1))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcb_$c -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile --pairs-total 0 > gpurun_out/pmcb_$c.log 2>&1
  echo "$c rc=$?"
done
python - <<PY
import csv, glob, collections, json
fam = collections.defaultdict(lambda: {"dispatches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob('gpurun_out/pmcb_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if k.startswith('void '): k = k[5:]
            if k.startswith('at::') or 'rocclr' in k or 'rocprim' in k or not k: continue
            fam[k][c + "_KB"] += float(r['Counter_Value'])
            if c == "FETCH_SIZE": fam[k]["dispatches"] += 1
pairs = 2   # --steps 1 --warmup 0: the timed step + the untimed per-family profiling step
sw = [v for k, v in fam.items() if 'k_sweep' in k]
launches = sum(v["dispatches"] for k, v in fam.items() if 'k_sweep2' in k)      # one launch = prep + main
fb = sum(v["FETCH_SIZE_KB"] for v in sw) * 1024 / max(launches, 1)
wb = sum(v["WRITE_SIZE_KB"] for v in sw) * 1024 / max(launches, 1)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) on \`python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile\` (one dense 9000x4000 pair per step; the run executes 2 pairs: the timed step + the untimed per-family profiling step). Values are KB summed over all dispatches of a kernel family. gfx950: FETCH_SIZE counts 1/2 of wide (16 B/lane) coalesced reads, so the true read side lies in [raw, 2*raw] (MI355X_MICROARCH.md, HBM section).",
       "pairs_in_run": pairs, "families": fam,
       "sweep_per_launch": {"launches": launches, "fetch_bytes_raw": fb, "write_bytes": wb, "traffic_bytes_lo": fb + wb, "traffic_bytes_hi": 2 * fb + wb},
       "per_pair": {"fetch_bytes_raw": sum(v["FETCH_SIZE_KB"] for v in fam.values()) * 1024 / pairs, "write_bytes": sum(v["WRITE_SIZE_KB"] for v in fam.values()) * 1024 / pairs}}
json.dump(out, open('gpurun_out/${R}_pmc_bench.json', 'w'), indent=1)
print(json.dumps(out["sweep_per_launch"]), json.dumps(out["per_pair"]))
PY
python - <<PY
import csv
rows = list(csv.DictReader(open('gpurun_out/${R}_bench_kernel_stats.csv')))
for r in rows[:12]:
    if 'at::' in r['Name']: continue
    print("%-60s calls %6s total_ms %10.3f avg_us %10.2f pct %6s" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
cat gpurun_out/${R}_bench_line.json | cut -c1-400
rm -rf gpurun_out/prof_$R gpurun_out/pmcb_*   # raw traces: tens of MB (gpurun copies back 64 MiB at most)
