#!/usr/bin/env python
"""Hardware-side floor of ONE step of the exact wavefront sweep (k_sweep2, compute_band): runs in the build container, no GPU.

  1. compiles csrc/kernels_sweep2.hip to gfx950 assembly with the product's flags (or takes --asm FILE),
  2. cuts the three steady-state loops of compute_band<TOP = 0 / 1 / 2> out of k_sweep2<TR=0, FWD=1, SPARSE=0, MODE=0> (each is the
     8-step unrolled chunk; a step starts at its first `row_newbcast:0` DPP move = the hand-over of the previous step's result),
  3. builds the register dependency graph of the not-taken (steady-state) path and prices it two ways:
       * recurrence  = the loop-carried dependency cycle alone: longest latency-weighted path through two copies of the chunk
                       minus the path through one, / 8 steps -- what an infinitely wide machine would need per step;
       * in-order    = one wave issuing the very instruction stream in order (issue cost per instruction, results available
                       after their latency, s_waitcnt lgkmcnt(0) drains the LDS queue) -- what ONE wave alone on a SIMD needs
                       per step; this is the floor bench.py reports as roofline.latency_bound.hw_floor_us,
  4. writes profiles/r05_sweep_step_isa.txt (annotated listing of one step with the critical path marked) and .json.

Latencies.  From /opt/skills/guides/MI355X_MICROARCH.md: wave64 VALU issue 2 cycles (SIMD-32), dependent VALU ~4 cycles,
ds_read issue->use ~50 cycles, DS cycles per wave-instruction (LDS table).  Where the guide is silent the numbers are this repo's own
micro-benchmarks on MI355X (tests/micro/issue_rate.hip): DPP move ~11 cycles to a dependent use, v_cmp -> select
~7, transcendental (v_sqrt_f32) 8 issue / ~16 result, taken branch ~30 (none on the steady-state path), v_readfirstlane -> SALU ~8.
Every such assumption is listed in the output so that the floor can be recomputed with other values (--lat name=cycles).
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "panorama-opticalflow_amd", "csrc", "kernels_sweep2.hip")
KERNEL = os.environ.get("ISA_KERNEL") or "_ZN2pf8k_sweep2INS_12_GLOBAL__N_16SwGeomILi4ELi1EEELb0ELb1ELb0ELi0EEE"      # k_sweep2<SwLatency, TR = false, FWD = true, SPARSE = false, MODE = 0> (ISA_KERNEL=... overrides: ...SwGeomILi8ELi2EEE... is the wide form)

# Two pricings.  "lone" (the default, = hw_floor_us): what ONE wave alone on a SIMD gets on MI355X, measured with
# tests/micro/lat_probe.hip (profiles/r03_lat_probe.txt): a wave is offered an issue slot only every ~5.5 cycles, whether or not the
# instruction depends on the previous one -- the guide's 2-cycle wave64 VALU rate needs at least two waves per SIMD, and the sweep's
# dependency chain lives in ONE wave per band.  "guide": MI355X_MICROARCH.md's throughput-side numbers (2-cycle issue, ~4-cycle
# dependent VALU, ~50-cycle ds_read), i.e. the same stream on an ideal single-wave pipeline; reported beside it.
# Round 6 (tests/micro/slot_model.py, profiles/r06_slot_model.txt): ONE wave alone on its SIMD issues one instruction per ~4.46 cycles WHATEVER it is --
# dependent or not, scalar, vector, DPP or an s_nop state (v_add chains 4.34-4.46 with 0..4 independent instructions between the links; "a dependent
# instruction costs 8.25" of round 5 was hipcc's own s_nop between two asm statements).  What costs more than a slot: consecutive packed fp32
# (5.3-5.5 each), v_rsq (+4), an LDS instruction (+6 on average among the step's nine: 249 instead of 195 cycles for 44 v_add + all nine), a
# not-taken branch (+4..+8).  Results are there for the next slot (no extra dependent latency) except: v_rsq -> reader one slot later (the
# assembly carries that wait state), LDS reads (ds_read2_b64 x2 -> first use 94 cycles, ds_read_b64 / b32 61, b128 ~70).
LAT_LONE = {"valu": 4.46, "pk": 5.0, "trans": 8.6, "dpp": 4.46, "cmp": 4.46, "ds_read": 52, "salu": 4.46, "readlane": 4.46}
ISSUE_LONE = {"valu": 4.46, "pk": 5.0, "trans": 8.0, "dpp": 4.46, "cmp": 4.46, "salu": 4.46, "readlane": 4.46, "nop": 4.46, "branch": 8.0, "wait": 1}
LDS_ISSUE_EXTRA = 6.0   # cycles an LDS instruction holds the wave's issue beyond a slot (lone-wave pricing only)
LAT_GUIDE = {"valu": 4, "pk": 4, "trans": 16, "dpp": 11, "cmp": 7, "ds_read": 50, "salu": 4, "readlane": 8}
ISSUE_GUIDE = {"valu": 2, "pk": 2, "trans": 8, "dpp": 2, "cmp": 2, "salu": 2, "readlane": 2, "nop": 1, "branch": 2, "wait": 1}
LAT = dict(LAT_LONE)
ISSUE = dict(ISSUE_LONE)
DS_CYC_LONE = {"ds_read_b32": 9, "ds_read_b64": 9, "ds_read_b128": 18, "ds_read2_b64": 38, "ds_read2_b32": 9}   # + LAT["ds_read"] = issue -> first use, lone wave (slot_model)
DS_CYC = {"ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read2st64_b64": 8, "ds_read2_b64": 8, "ds_read2_b32": 4,
          "ds_write_b32": 4, "ds_write_b64": 6, "ds_write_b128": 13}   # guide, LDS table: cycles per wave-instruction


def compile_asm():
    """the assembly that ships: hipcc -S with the Makefile's flags, then tools/asm_sched.py with the Makefile's SCHED_FLAGS"""
    out, sched_out = "/tmp/isa_chain_sweep2.s", "/tmp/isa_chain_sweep2.sched.s"
    pkg = os.path.join(ROOT, "panorama-opticalflow_amd")
    mk = open(os.path.join(pkg, "Makefile")).read()
    var = lambda name: re.search(r"^%s\s*\??=\s*(.*)$" % re.escape(name), mk, re.M).group(1).strip()
    # ISA_FLAGS="..." replaces the scheduler flags for what-if runs
    sched = os.environ.get("ISA_FLAGS", var("FLAGS_kernels_sweep2")).split()
    subprocess.check_call([var("HIPCC")] + var("FLAGS").replace("$(ARCH)", var("ARCH")).split() + sched + ["-S", "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
    if os.environ.get("ISA_NO_SCHED"):
        return out
    subprocess.check_call([sys.executable, os.path.join(pkg, "tools", "asm_sched.py"), out, sched_out] + var("SCHED_FLAGS").split(), stderr=subprocess.DEVNULL)
    return sched_out


REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]")


def regs_of(op):
    op = op.strip()
    out = []
    if op in ("vcc", "exec", "scc", "m0"):
        return [op]
    for m in REG.finditer(op):
        if m.group(1):
            out.append(m.group(1) + m.group(2))
        else:
            out += [m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
    if "vcc" in op.split("|")[0].split()[0:1]:
        out.append("vcc")
    return out


def split_ops(s):
    ops, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur)
    return [o.strip() for o in ops]


class Ins:
    __slots__ = ("text", "mn", "kind", "dst", "src", "lat", "issue", "lds")


def classify(line):
    t = line.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        return None
    t = t.split(";")[0].strip()
    if not t:
        return None
    parts = t.split(None, 1)
    mn = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    # strip DPP / modifier tails from the operand list
    mods = ""
    m = re.search(r"\s(row_|quad_perm|neg_lo|neg_hi|op_sel|offset|bound_ctrl|clamp|bank_mask|row_mask)", " " + rest)
    if m:
        mods = rest[m.start():]
        rest = rest[:m.start()]
    ops = split_ops(rest)
    i = Ins(); i.text = t; i.mn = mn; i.lds = False
    nodst = mn.startswith(("ds_write", "ds_add", "ds_max", "ds_min", "ds_or", "global_store", "buffer_store", "s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_sleep", "s_barrier", "s_setprio", "s_endpgm", "s_sethalt"))
    i.dst, i.src = [], []
    if mn.startswith("s_cmp"):
        i.dst = ["scc"]; i.src = sum((regs_of(o) for o in ops), [])
    elif mn.startswith("s_cbranch_scc"):
        i.src = ["scc"]
    elif mn.startswith("s_cbranch_vcc"):
        i.src = ["vcc"]
    elif mn.startswith("s_cbranch_exec"):
        i.src = []
    elif nodst:
        i.src = sum((regs_of(o) for o in ops), [])
    else:
        i.dst = regs_of(ops[0]) if ops else []
        i.src = sum((regs_of(o) for o in ops[1:]), [])
        if mn.endswith("_dpp") and "bound_ctrl" not in mods:
            i.src += i.dst          # lanes the move does not write keep the old value
        if "saveexec" in mn:
            i.dst = i.dst + ["exec"]
    if mn.startswith("ds_read"):
        i.kind = "ds_read"; i.lds = True
    elif mn.startswith("ds_"):           # stores and LDS atomics without a returned value
        i.kind = "ds_write"; i.lds = True
    elif mn == "s_waitcnt":
        i.kind = "wait"
    elif mn == "s_nop":
        i.kind = "nop"
    elif mn.startswith(("s_cbranch", "s_branch")):
        i.kind = "branch"
    elif mn.startswith("v_readfirstlane") or mn.startswith("v_readlane"):
        i.kind = "readlane"
    elif mn.startswith("s_"):
        i.kind = "salu"
    elif mn.endswith("_dpp"):
        i.kind = "dpp"
    elif mn.startswith("v_cmp"):
        i.kind = "cmp"
    elif mn.startswith(("v_sqrt", "v_rcp", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        i.kind = "trans"
    elif mn.startswith("v_pk_"):
        i.kind = "pk"
    else:
        i.kind = "valu"
    return i


def price(i, lat, issue):
    if i.kind == "ds_read":
        base = i.mn.split()[0]
        lone = issue is not ISSUE_GUIDE
        i.issue = issue["valu"] + (LDS_ISSUE_EXTRA if lone else 0); i.lat = lat["ds_read"] + (DS_CYC_LONE if lone else DS_CYC).get(base, 2)
    elif i.kind == "ds_write":
        lone = issue is not ISSUE_GUIDE
        i.issue = (issue["valu"] + LDS_ISSUE_EXTRA) if lone else max(issue["valu"], DS_CYC.get(i.mn, 4)); i.lat = 0
    elif i.kind == "nop":
        i.issue = issue["nop"] * (int(i.text.split()[1], 0) + 1); i.lat = 0     # s_nop N = N + 1 wait states
    elif i.kind in ("wait", "branch"):
        i.issue = issue[i.kind]; i.lat = 0
    else:
        i.issue = issue[i.kind]; i.lat = lat[i.kind]


def steady_path(lines, lo, hi):
    """instructions of lines[lo:hi] on the fall-through path: blocks reached only by a taken forward branch are not in the range
    (the compiler moved the cold paths -- window fall-back, IEEE redo, producer edge -- behind the loop)"""
    out = []
    for ln in lines[lo:hi]:
        i = classify(ln)
        if i is not None:
            out.append(i)
    return out


def simulate(ins, copies, lat):
    """one wave, in order: returns the finish time of every copy of the chunk"""
    ready = {}            # register -> time its value is available
    t = 0.0
    lds_pending = []      # completion times of outstanding LDS reads
    ends = []
    crit_parent = {}
    times = []
    for c in range(copies):
        for k, i in enumerate(ins):
            start = t
            why = None
            for r in i.src:
                if ready.get(r, (0.0, None))[0] > start:
                    start = ready[r][0]; why = ready[r][1]
            if i.kind == "wait" and "lgkmcnt(0)" in i.text and lds_pending:
                m = max(lds_pending)
                if m > start:
                    start = m; why = "lds"
                lds_pending = []
            done = start + i.lat
            for r in i.dst:
                ready[r] = (done, (c, k))
            if i.kind == "ds_read":
                lds_pending.append(done)
            times.append((c, k, start, why))
            t = start + i.issue
        ends.append(t)
    return ends, times


def longest_path(ins, copies, lat):
    """pure data-dependency longest path (latency-weighted, no issue limits) through `copies` copies of the chunk"""
    ready = {}
    best = 0.0
    for c in range(copies):
        for i in ins:
            s = 0.0
            for r in i.src:
                s = max(s, ready.get(r, 0.0))
            d = s + max(i.lat, 0)
            for r in i.dst:
                ready[r] = d
            best = max(best, d)
    return best, ready


def analyse(body, loops, lat, issue, clock):
    out = []
    for li, lm in enumerate(loops):
        step_len = lm[1] - lm[0]
        lo, hi = lm[0], lm[7] + step_len       # 8 steps, the last one assumed as long as the others (its tail is the loop back-edge code)
        # extend to the loop's backward branch so that the chunk-level bookkeeping is included once per 8 steps
        for n in range(lm[7], min(len(body), lm[7] + 400)):
            if re.match(r"\s*s_cbranch_\w+\s+\.LBB", body[n]) and any(body[m].startswith(body[n].split()[-1] + ":") for m in range(max(0, lm[0] - 400), lm[0] + 1)):
                hi = n + 1; break
        ins = steady_path(body, lo, hi)
        for i in ins:
            price(i, lat, issue)
        nvalu = sum(1 for i in ins if i.kind in ("valu", "pk", "cmp", "dpp", "trans", "readlane"))
        nlds = sum(1 for i in ins if i.lds)
        nsalu = sum(1 for i in ins if i.kind in ("salu", "branch", "wait", "nop"))
        ends, times = simulate(ins, 6, lat)
        per_chunk = ends[-1] - ends[-2]
        lp2, _ = longest_path(ins, 3, lat)
        lp1, _ = longest_path(ins, 2, lat)
        has_top = any("ds_read_b64" in i.text for i in ins) and sum(1 for i in ins if i.mn == "ds_read_b32") >= 8
        e = {"loop": li, "instructions_per_8_steps": len(ins),
             "per_step": {"instructions": round(len(ins) / 8, 1), "vector": round(nvalu / 8, 1), "lds": round(nlds / 8, 1), "scalar_and_control": round(nsalu / 8, 1)},
             "reads_top_neighbour_from_lds": bool(has_top),
             "issue_slots_per_step": round(sum((int(i.text.split()[1], 0) + 1) if i.kind == "nop" else (0 if i.kind == "wait" else 1) for i in ins) / 8, 1),
             "issue_only_cycles_per_step": round(sum(i.issue for i in ins) / 8, 1), "recurrence_cycles_per_step": round((lp2 - lp1) / 8, 1),
             "in_order_cycles_per_step": round(per_chunk / 8, 1), "in_order_us_per_step": round(per_chunk / 8 / (clock * 1e3), 4)}
        out.append((e, ins, times))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--lat", action="append", default=[], help="override a lone-wave latency: name=cycles (valu, pk, trans, dpp, cmp, ds_read, salu, readlane)")
    ap.add_argument("--issue", action="append", default=[], help="override a lone-wave issue cost: name=cycles (valu, pk, trans, dpp, cmp, salu, readlane)")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_sweep_step_isa"))
    a = ap.parse_args()
    lat, issue = dict(LAT_LONE), dict(ISSUE_LONE)
    for kv in a.lat:
        k, v = kv.split("="); lat[k] = float(v)
    for kv in a.issue:
        k, v = kv.split("="); issue[k] = float(v)
    path = a.asm or compile_asm()
    lines = open(path).read().split("\n")
    k0 = next(n for n, l in enumerate(lines) if l.startswith(KERNEL))
    k1 = next(n for n in range(k0 + 1, len(lines)) if lines[n].strip().startswith("s_endpgm"))
    body = lines[k0:k1]
    marks = [n for n, l in enumerate(body) if "row_newbcast:0" in l]
    if any("v_mov_b64_dpp" in body[n] for n in marks):
        marks = [n for n in marks if "v_mov_b64_dpp" in body[n]]   # round 6: ONE 64-bit move per step
    else:
        marks = marks[::2]                   # two moves (x, y) per step
    loops = [marks[i:i + 8] for i in range(0, len(marks) - 7, 8)]
    lone = analyse(body, loops, lat, issue, a.clock_ghz)
    guide = analyse(body, loops, LAT_GUIDE, ISSUE_GUIDE, a.clock_ghz)
    pick = lambda rs, key: max((e[key] for e, _, _ in rs if e["reads_top_neighbour_from_lds"]), default=max(e[key] for e, _, _ in rs))
    result = {"kernel": "pf::k_sweep2<false, true, false, 0> (normal orientation, forward, dense, records from k_sweep_prep)", "clock_ghz": a.clock_ghz,
              "lone_wave": {"latencies_cycles": lat, "issue_cycles": issue, "lds_issue_extra_cycles": LDS_ISSUE_EXTRA, "source": "tests/micro/slot_model.py on MI355X (profiles/r06_slot_model.txt)", "loops": [e for e, _, _ in lone]},
              "guide_pipeline": {"latencies_cycles": LAT_GUIDE, "issue_cycles": ISSUE_GUIDE, "source": "MI355X_MICROARCH.md (2-cycle wave64 VALU, ~4-cycle dependent VALU, ~50-cycle ds_read; DPP / cmp / sqrt from tests/micro/issue_rate.hip)",
                                 "loops": [e for e, _, _ in guide]},
              "ds_cycles": DS_CYC,
              "isa_model_us": pick(lone, "in_order_us_per_step"), "isa_model_cycles": pick(lone, "in_order_cycles_per_step"),
              "issue_slots_per_step": pick(lone, "issue_slots_per_step"),
              "recurrence_floor_cycles": pick(lone, "recurrence_cycles_per_step"),
              "guide_pipeline_floor_us": pick(guide, "in_order_us_per_step"), "guide_pipeline_floor_cycles": pick(guide, "in_order_cycles_per_step"),
              "how_to_recompute": "python tests/micro/isa_chain.py [--lat name=cycles ...] [--issue name=cycles ...]   (compiles csrc/kernels_sweep2.hip with the product's flags; no GPU needed)"}
    json.dump(result, open(a.out + ".json", "w"), indent=1)
    with open(a.out + ".txt", "w") as f:
        f.write("Steady-state step of the exact wavefront sweep -- k_sweep2<TR=0, FWD=1, SPARSE=0, MODE=0>, compute_band (CPU/PixFlow.hpp:315-324,\n"
                "342-386, 427-456: gate, proposeFlowUpdate from L and T, errorGradient, flow -= 0.5 * grad; six errorFunction evaluations in six lanes).\n"
                "Generated by tests/micro/isa_chain.py from `hipcc -S` of csrc/kernels_sweep2.hip with the product's flags; recompute with\n"
                "`python tests/micro/isa_chain.py [--lat name=cycles] [--issue name=cycles]`.\n\n"
                "WHAT BOUNDS A STEP.  The sweep's dependency chain lives in ONE wave per band of 8 rows, and on gfx950 one wave alone on its SIMD issues ONE\n"
                "instruction per ~4.46 cycles whatever it is -- dependent or not, vector, scalar, DPP, or an s_nop state (tests/micro/slot_model.py,\n"
                "profiles/r06_slot_model.txt; the guide's 2-cycle wave64 VALU rate needs two or more waves per SIMD).  So a step costs its issue SLOTS\n"
                "(instructions + wait states) x that interval, plus what a few instruction kinds hold the issue longer (consecutive packed fp32, v_rsq,\n"
                "LDS instructions, branches) and whatever LDS latency the stream does not cover; the loop-carried dependency cycle (`recurrence`) is\n"
                "shorter and not the limit.  This is the PRODUCT's assembly: hipcc's output after tools/asm_sched.py (Makefile SCHED=1).\n\n")
        f.write("Three instances of the loop exist in the kernel (compute_band<TOP>: first band of a sweep / band inside a workgroup / first band of a\n"
                "workgroup); each is the 8-step unrolled chunk (a step starts at its first `row_newbcast:0` DPP move).\n\n")
        for (e, _, _), (g, _, _) in zip(lone, guide):
            f.write("  loop %d: %.1f instructions per step (%.1f vector, %.1f LDS, %.1f scalar/control)%s\n"
                    "      lone-wave pricing : issue slots alone %4.0f cycles | recurrence %4.0f | ONE wave in order %4.0f cycles = %.4f us @ %.1f GHz\n"
                    "      guide pipeline    : issue slots alone %4.0f cycles | recurrence %4.0f | ONE wave in order %4.0f cycles = %.4f us\n"
                    % (e["loop"], e["per_step"]["instructions"], e["per_step"]["vector"], e["per_step"]["lds"], e["per_step"]["scalar_and_control"],
                       ", row 0 takes its top neighbour from the LDS ring" if e["reads_top_neighbour_from_lds"] else ", no top neighbour (first band)",
                       e["issue_only_cycles_per_step"], e["recurrence_cycles_per_step"], e["in_order_cycles_per_step"], e["in_order_us_per_step"], a.clock_ghz,
                       g["issue_only_cycles_per_step"], g["recurrence_cycles_per_step"], g["in_order_cycles_per_step"], g["in_order_us_per_step"]))
        f.write("\nisa_model_us = %.4f (%.1f issue slots per step; lone-wave pricing, max over the loops that read a top neighbour; bench.py:\n"
                "roofline.latency_bound.isa_model_us).  It is a MODEL of one wave in order with this round's measured slot prices, to be read beside the\n"
                "measured step of a lone band (bench.py t_step_us, HIP events around a whole 8 x 4096 launch, i.e. including the launch's start-up and the\n"
                "loaders' first round trips): the measurement must not fall below it.  On an ideal single-wave pipeline (guide numbers: 2-cycle issue) the\n"
                "same stream would need %.4f us: the factor between the two is the lone-wave issue interval, which no instruction placement changes --\n"
                "only fewer slots per step do (round 1: 0.70 us -> 2: 0.35 -> 3: 0.275 -> 5: 0.267 -> 6: see bench; DESIGN.md 3.3), or a second wave that\n"
                "shares the step's work, which the step's own dependency chain forbids (every hand-over between waves goes through LDS: >= 60 cycles per hop).\n\n"
                % (result["isa_model_us"], result["issue_slots_per_step"], result["guide_pipeline_floor_us"]))
        # annotated listing of the slowest top-reading loop, lone-wave pricing
        e, ins, times = max((r for r in lone if r[0]["reads_top_neighbour_from_lds"]), key=lambda r: r[0]["in_order_cycles_per_step"], default=lone[0])
        n = len(ins)
        tl = {(c, k): (start, why) for (c, k, start, why) in times}
        crit = set()
        cur = (5, n - 1)
        guard = 0
        while cur and guard < 20000:
            guard += 1
            crit.add(cur)
            start, why = tl[cur]
            c, k = cur
            if why is None:
                cur = (c, k - 1) if k > 0 else ((c - 1, n - 1) if c > 0 else None)   # issue-order predecessor
            elif why == "lds":
                cand = [(cc, kk) for (cc, kk, st, w) in times if (cc, kk) < cur and ins[kk].kind == "ds_read"]
                cur = max(cand, key=lambda q: tl[q][0] + ins[q[1]].lat) if cand else None
            else:
                cur = why
        f.write("Loop %d, chunk in steady state (6th simulated chunk), lone-wave pricing: issue time (cycles since the chunk's first instruction),\n"
                "instruction.  '*' = on the critical chain of the in-order model (walked back from the chunk's last instruction through the\n"
                "latest-arriving operand, or the issue-order predecessor when no operand was late -- i.e. when the ISSUE INTERVAL, not a dependency, set\n"
                "the time); '>' = waited for an operand; 'L' = waited for the LDS queue (s_waitcnt lgkmcnt(0)).\n\n" % e["loop"])
        base = tl[(5, 0)][0]
        step_no = 0
        nstar = nwait = 0
        for k, i in enumerate(ins):
            if "row_newbcast:0" in i.text and (i.mn == "v_mov_b64_dpp" or k == 0 or "row_newbcast:0" not in ins[k - 1].text):
                step_no += 1
                f.write("  ---- step %d of the chunk ----\n" % step_no)
            start, why = tl[(5, k)]
            nstar += (5, k) in crit; nwait += why is not None
            flag = ("*" if (5, k) in crit else " ") + ("L" if why == "lds" else (">" if why is not None else " "))
            f.write("  %s %7.0f  %s\n" % (flag, start - base, i.text))
        f.write("\n%d of the chunk's %d instructions issued the moment their slot came (no operand wait); %d waited for an operand or the LDS queue.\n" % (n - nwait, n, nwait))
    print(json.dumps({k: result[k] for k in ("isa_model_us", "isa_model_cycles", "issue_slots_per_step", "recurrence_floor_cycles", "guide_pipeline_floor_us")}))


if __name__ == "__main__":
    main()
