#!/bin/bash
# Round 6 (review item 3): hardware counters under the lone pair's sweep.  One rocprofv3 --pmc pass per counter group (with --kernel-trace only;
# never combined with sys/hip/hsa tracing) of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile --pairs-total 0`
# (one dense 9000x4000 pair per step; the run executes the timed step only), summed per kernel family over the k_sweep2 / k_sweep_prep dispatches.
#   -> gpurun_out/<R>_sweep_sq.json      (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${1:-r06}
D=gpurun_out/sq_$R; rm -rf $D; mkdir -p $D
rocprofv3 -L > $D/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*" $D/avail.txt | sort -u > gpurun_out/${R}_sq_counters_available.txt
G=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  G=$((G+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $D/g$G -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile --pairs-total 0 > $D/g$G.log 2>&1
  echo "group $G ($set) rc=$?"
done
D=$D R=$R python - <<'PY'
import csv, glob, collections, json, os
D = os.environ['D']; R = os.environ['R']
fam = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(int)
seen_first = None
for f in sorted(glob.glob(D + '/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'pf::' not in n: continue
        n = n[n.index('pf::') + 4:].split('(')[0]
        if 'k_sweep' not in n: n = n.split('<')[0]
        fam[n][r['Counter_Name']] += float(r['Counter_Value'])
        if seen_first is None: seen_first = r['Counter_Name']
        if r['Counter_Name'] == seen_first: disp[n] += 1
out = {"note": "rocprofv3 --pmc (one pass per group of 4 SQ counters, --kernel-trace only) on `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-profile --pairs-total 0`: ONE dense 9000x4000 pair, both directions. Counter values summed over all dispatches of a kernel family and over all XCDs / SEs as rocprofv3 reports them. SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles) per MI355X_MICROARCH.md. A k_sweep2 workgroup holds 4 compute waves + 7 helper waves (loaders, publisher, poller, drainer): per-kernel counters cover all 11.",
       "dispatches": disp, "families": fam}
sw = collections.defaultdict(float)
for k, v in fam.items():
    if k.startswith('k_sweep2'):
        for c, x in v.items(): sw[c] += x
if sw:
    d = {"counters": dict(sw)}
    g = lambda c: sw.get(c, 0.0)
    if g("SQ_INSTS_VALU"):
        d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / max(g("SQ_WAVES"), 1)
        d["active_valu_quadcycles_per_valu_inst"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
        d["active_valu_cycles_per_valu_inst"] = 4 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
    if g("SQ_WAVE_CYCLES"):
        allinst = g("SQ_INSTS_VALU") + g("SQ_INSTS_SALU") + g("SQ_INSTS_LDS") + g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR") + g("SQ_INSTS_SMEM")
        d["wave_cycles_per_instruction_all_waves"] = 4 * g("SQ_WAVE_CYCLES") / max(allinst, 1)
        d["share_wait_any"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        d["share_wait_inst_any"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        d["share_active_inst_any"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
        d["share_wait_inst_lds_of_wait_inst_any"] = g("SQ_WAIT_INST_LDS") / max(g("SQ_WAIT_INST_ANY"), 1)
    if g("SQ_LDS_IDX_ACTIVE"): d["lds_bank_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    out["k_sweep2_all"] = d
json.dump(out, open('gpurun_out/%s_sweep_sq.json' % R, 'w'), indent=1)
print(json.dumps(out.get("k_sweep2_all", {}), indent=1)[:3000])
PY
for f in $D/g*.log; do tail -n 2 $f | cut -c1-200; done
rm -rf $D
