#!/usr/bin/env python3
"""Post-pass over hipcc's gfx950 assembly of csrc/kernels_sweep2.hip: takes the WAIT STATES out of the sweep kernels' straight-line blocks.

Why.  One wave alone on its SIMD -- the regime the latency-form sweep lives in -- issues ONE instruction every 4 cycles,
whatever it is and whatever it depends on (tests/micro/slot_model.py, profiles/r06_slot_model.txt: a dependent v_add chain runs at 4.25 cycles
per instruction, like independent ones; an `s_nop 0` costs the same 4 cycles as an instruction).  So a step costs its issue SLOTS, and hipcc
spends 5-8 of a step's ~140 on s_nop:
  * false positives: ROCm 7.2's LLVM separates every v_pk_*_f32 (and every inline-asm statement) from an instruction that reads its result
    (its dst_sel hazard test misreads op_sel_hi of VOP3P; the hazard is about partial register writes, a packed fp32 result is two whole
    registers -- csrc/exact_forms.hpp, tests/micro/pk_hazard_probe.hip, and k_pk_probe at load time vouch for that on the device itself);
  * real hazards (a DPP read of a VGPR a VALU instruction wrote <= 2 slots ago, a VALU read of a v_rsq result 1 slot later, a VALU read of an
    SGPR / VCC a VALU instruction wrote <= 2 slots ago) that an independent instruction from further down could fill just as well.
The compiler cannot be told either (asm blocks only move its wait states: DESIGN.md 3.3).  This pass re-derives the wait states a block needs
from its own hazard table, lets later INDEPENDENT instructions of the same block fill them (list scheduling with the compiler's order as the
priority: an instruction only moves up when the one in front of it must wait), and emits an s_nop only where nothing else can go.

What it touches: basic blocks (label / branch delimited) of the kernels whose name matches --kernels, and only blocks made of instructions
it knows (see KNOWN); a block with anything else -- exec writes, memory instructions other than LDS, messages, barriers -- is left as hipcc wrote
it.  Instructions never cross a block boundary, LDS instructions keep their order among themselves and relative to every s_waitcnt, and an
instruction that touches a register any LDS read of the kernel loads never crosses an s_waitcnt.  Every rescheduled block is re-checked: same
instructions, every register dependency (RAW / WAR / WAW, incl. SCC / VCC / EXEC / M0) in its old order, every hazard of the table satisfied.

Hazard table (wait states = instructions or s_nop states between producer and consumer), gfx950; lower bounds taken from LLVM's
GCNHazardRecognizer and cross-checked against the minimum distances hipcc itself leaves anywhere in the 200k-line listing of this file:
  VALU (any: also packed, transcendental, DPP) writes a VGPR  -> DPP instruction reads it as its SOURCE: 2
                                                              -> DPP move keeps it as the OLD value of the lanes its masks leave out: --dpp-old-wait (2 = LLVM's rule)
  transcendental (v_rsq / v_sqrt / v_rcp ...) writes a VGPR   -> any VALU instruction reads it: 1
  VALU writes an SGPR / VCC (v_cmp, v_readfirstlane, ...)     -> VALU instruction reads it (mask or operand): 2
  VALU writes a VGPR                                          -> v_readfirstlane / v_readlane reads it: 1;  v_permlane* reads it: 2
  SALU writes M0                                              -> an instruction that uses M0 implicitly (LDS-DMA): 1
  a value that enters the block (producer unknown)            -> the consumer keeps at least min(its old distance from the block's start, the rule)
  packed fp32 result -> dependent VALU: 0 (see above; --pk-wait N restores the compiler's belief for an A/B)

usage: asm_sched.py in.s out.s [--kernels REGEX] [--pk-wait N] [--report FILE] [--no-move]
"""
import argparse
import re
import sys

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\bvcc(?:_lo|_hi)?\b|\bexec(?:_lo|_hi)?\b|\bm0\b|\bscc\b")
MODS = re.compile(r"\s(row_|quad_perm|neg_lo|neg_hi|op_sel|offset|bound_ctrl|clamp|bank_mask|row_mask|wave_|mul:|div:|sc0|sc1|nt\b|gds|dst_sel|src0_sel|src1_sel)")

# mnemonic prefixes this pass understands (operand 0 = destination unless listed in NODST); anything else => the block is left alone
KNOWN_VALU = ("v_add_", "v_sub_", "v_subrev_", "v_mul_", "v_fma_", "v_fmac_", "v_mad_u32_u24", "v_mad_i32_i24", "v_max_", "v_min_", "v_max3_", "v_min3_", "v_med3_", "v_and_", "v_or_", "v_xor_", "v_not_",
              "v_lshl", "v_lshr", "v_ashr", "v_lshl_add_u32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_add3_u32", "v_add_lshl_u32", "v_xad_u32", "v_bfe_", "v_bfi_", "v_perm_b32", "v_alignbit_", "v_mov_b32", "v_mov_b64", "v_cndmask_b32",
              "v_cvt_", "v_fract_", "v_frexp_", "v_ldexp_", "v_floor_", "v_ceil_", "v_trunc_", "v_rndne_", "v_pk_", "v_cmp_", "v_rsq_", "v_sqrt_", "v_rcp_", "v_readfirstlane_b32", "v_accvgpr_", "v_sad_", "v_mbcnt_")
KNOWN_SALU = ("s_add_", "s_sub_", "s_mul_i32", "s_mulk_i32", "s_and_b", "s_or_b", "s_xor_b", "s_andn2_b", "s_orn2_b", "s_not_b", "s_lshl_b", "s_lshr_b", "s_ashr_", "s_mov_b", "s_movk_i32", "s_cmp_", "s_cmpk_", "s_cselect_b", "s_min_", "s_max_", "s_bfe_",
              "s_addk_i32", "s_lshl1_add_u32", "s_lshl2_add_u32", "s_lshl3_add_u32", "s_lshl4_add_u32", "s_bitcmp", "s_ff1_", "s_flbit_", "s_bcnt", "s_sext_", "s_abs_", "s_addc_u32", "s_subb_u32", "s_mul_hi_")
KNOWN_LDS = ("ds_read", "ds_write")
KNOWN_MISC = ("s_waitcnt", "s_nop", "s_cbranch_", "s_branch")
NO_SCC = ("s_mov_b", "s_movk_i32", "s_mul_i32", "s_mulk_i32", "s_cselect_b", "s_mul_hi_", "s_sext_", "s_cmov")     # SALU that leaves SCC alone; everything else is taken to write it
TRANS = ("v_rsq_", "v_sqrt_", "v_rcp_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def regs_of(op):
    out = []
    for m in REG.finditer(op):
        g = m.group(0)
        if g.startswith("vcc"):
            out.append("vcc")
        elif g.startswith("exec"):
            out.append("exec")
        elif g in ("m0", "scc"):
            out.append(g)
        elif m.group(1):
            out.append(m.group(1) + m.group(2))
        else:
            out += [m.group(3) + str(i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
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
    __slots__ = ("text", "mn", "defs", "uses", "kind", "known", "dpp_reads", "dpp_old", "nop_states", "orig", "lds", "sgpr_by_valu", "is_branch")

    def __repr__(self):
        return self.text


def parse(text):
    """text: one instruction (no comment).  Returns an Ins; .known = False if this pass must not move anything around it."""
    i = Ins()
    i.text = text; i.known = True; i.dpp_reads = set(); i.dpp_old = set(); i.nop_states = 0; i.lds = False; i.sgpr_by_valu = False; i.is_branch = False
    parts = text.split(None, 1)
    mn = i.mn = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    m = MODS.search(" " + rest)
    if m:
        rest = rest[:m.start()]
    ops = split_ops(rest)
    defs, uses = set(), set()
    if mn == "s_nop":
        i.kind = "nop"; i.nop_states = int(ops[0], 0) + 1
    elif mn == "s_waitcnt":
        i.kind = "wait"; defs |= {"LDS", "VMEM", "LDSREGS"}; uses |= {"LDS", "VMEM", "LDSREGS"}
    elif mn.startswith(("s_cbranch_", "s_branch")):
        i.kind = "branch"; i.is_branch = True
        if "vcc" in mn: uses.add("vcc")
        if "scc" in mn: uses.add("scc")
        if "exec" in mn: uses.add("exec")
    elif mn.startswith("ds_read"):
        i.kind = "lds"; i.lds = True
        defs |= set(regs_of(ops[0])) | {"LDS"}; uses |= {"LDS"}
        for o in ops[1:]: uses |= set(regs_of(o))
    elif mn.startswith("ds_write"):
        i.kind = "lds"; i.lds = True
        defs |= {"LDS"}; uses |= {"LDS"}
        for o in ops: uses |= set(regs_of(o))
    elif mn.startswith("global_load_lds_"):
        # LDS-DMA (the throughput form's record stream): global memory -> LDS at M0's address; completion is counted in vmcnt.  Ordered with every
        # other memory instruction and every s_waitcnt through both pseudo registers; M0 is an implicit source (1 wait state behind an SALU write)
        i.kind = "vmem"
        defs |= {"LDS", "VMEM"}; uses |= {"LDS", "VMEM", "m0"}
        for o in ops: uses |= set(regs_of(o))
    elif mn.startswith("v_permlane32_swap") or mn.startswith("v_permlane16_swap"):
        i.kind = "valu"
        both = set()
        for o in ops: both |= set(regs_of(o))
        defs |= both; uses |= both
    elif mn.startswith(KNOWN_SALU):
        i.kind = "salu"
        if mn.startswith(("s_cmp_", "s_cmpk_", "s_bitcmp")):
            defs.add("scc")
            for o in ops: uses |= set(regs_of(o))
        else:
            defs |= set(regs_of(ops[0]))
            for o in ops[1:]: uses |= set(regs_of(o))
            if not mn.startswith(NO_SCC): defs.add("scc")
            if mn.startswith(("s_cselect_b", "s_addc_u32", "s_subb_u32", "s_cmov")): uses.add("scc")
            if mn.startswith(("s_addk_i32", "s_mulk_i32", "s_cmov")): uses |= set(regs_of(ops[0]))
        if "saveexec" in mn or "exec" in defs: i.known = False
    elif mn.startswith(KNOWN_VALU) and not mn.startswith("v_cmpx"):
        i.kind = "trans" if mn.startswith(TRANS) else "valu"
        d = set(regs_of(ops[0])) if ops else set()
        defs |= d
        for o in ops[1:]: uses |= set(regs_of(o))
        if mn.startswith(("v_fmac_", "v_mac_", "v_pk_fmac")): uses |= d
        if "dpp" in mn:
            i.kind = "dpp"
            src = set(r for r in uses if r[0] == "v")   # read through the cross-lane network: the DPP hazard proper
            uses |= d                                   # lanes a DPP move does not write keep the old value (bank / row masks, no bound_ctrl)
            i.dpp_reads = src
            i.dpp_old = set(r for r in d if r[0] == "v") - src
        if mn.startswith(("v_cmp_", "v_readfirstlane", "v_readlane")): i.sgpr_by_valu = True
        if mn.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_mad_u64", "v_mad_i64", "v_div_")): i.known = False   # second (scalar) destination: not modelled
        if "exec" in defs: i.known = False
    else:
        i.kind = "other"; i.known = False
    i.defs, i.uses = defs, uses
    return i


def depends(a, b):
    """b (later in program order) must stay behind a"""
    return bool(a.defs & (b.uses | b.defs)) or bool(a.uses & b.defs)


def hazard_need(prod, cons, reg, pk_wait):
    """wait states the table demands between `prod` (writes reg) and `cons` (reads reg)"""
    need = 0
    if reg[0] == "v":
        if prod.kind in ("valu", "trans", "dpp"):
            if reg in cons.dpp_reads: need = max(need, 2)
            if reg in cons.dpp_old: need = max(need, DPP_OLD_WAIT)
            if cons.mn.startswith(("v_readfirstlane", "v_readlane")): need = max(need, 1)
            if cons.mn.startswith("v_permlane"): need = max(need, 2)
        if prod.kind == "trans" and cons.kind in ("valu", "trans", "dpp"): need = max(need, 1)
        if pk_wait and prod.mn.startswith("v_pk_") and cons.kind in ("valu", "trans", "dpp"): need = max(need, pk_wait)
    elif reg[0] == "s" or reg == "vcc":
        if prod.sgpr_by_valu and cons.kind in ("valu", "trans", "dpp"): need = max(need, 2)
    elif reg == "m0":
        if prod.kind == "salu" and "_lds_" in cons.mn: need = max(need, 1)
    return need


DPP_OLD_WAIT = 2   # --dpp-old-wait: wait states between a VALU write of a register and a DPP move that only keeps it as the OLD value of masked-off lanes
HAZ_READ_MAX = 2   # the largest entry of the table: what a value entering the block may still need


def schedule_block(ins_with_nops, ldsregs, pk_wait, allow_move):
    """ins_with_nops: the block's instructions in hipcc's order (s_nop included, branch last if any).  Returns (new list of Ins incl. fresh nops, stats)."""
    term = ins_with_nops[-1] if ins_with_nops[-1].is_branch else None
    body = ins_with_nops[:-1] if term else list(ins_with_nops)
    # old wait-state position of every instruction (nops count their states)
    pos, seq = 0, []
    for i in body:
        if i.kind == "nop":
            pos += i.nop_states; continue
        i.orig = pos; pos += 1; seq.append(i)
    if term is not None:
        term.orig = pos
    old_slots = pos
    n = len(seq)
    for i in seq:   # anything that touches an LDS-loaded register stays on its side of every s_waitcnt
        if i.kind != "wait" and ((i.defs | i.uses) & ldsregs):
            i.uses = i.uses | {"LDSREGS"}
    preds = [[] for _ in range(n)]
    for b in range(n):
        for a in range(b):
            if depends(seq[a], seq[b]): preds[b].append(a)
    # values that enter the block: the consumer may not come closer to the block's start than it was (capped by the table's maximum)
    first_writer = {}
    entry_min = [0] * n
    for k, i in enumerate(seq):
        hz = set()
        if i.dpp_reads: hz |= i.dpp_reads
        if i.dpp_old and DPP_OLD_WAIT: hz |= i.dpp_old
        if i.kind in ("valu", "trans", "dpp"): hz |= set(r for r in i.uses if r[0] in "vs" or r == "vcc")
        if "_lds_" in i.mn: hz.add("m0")
        if any(r not in first_writer for r in hz): entry_min[k] = min(i.orig, HAZ_READ_MAX)
        for r in i.defs: first_writer.setdefault(r, k)
    done, order, slot_of = [False] * n, [], {}
    last_writer = {}          # register -> index in seq of its latest writer already emitted
    out, slot, nops_new = [], 0, 0
    remaining = n
    def ok_at(k, slot):
        i = seq[k]
        if slot < entry_min[k]: return False
        for r in i.uses:
            w = last_writer.get(r)
            if w is None: continue
            need = hazard_need(seq[w], i, r, pk_wait)
            if need and slot - slot_of[w] - 1 < need: return False
        return True
    while remaining:
        pick = None
        for k in range(n):
            if done[k]: continue
            if any(not done[p] for p in preds[k]):
                if not allow_move: break
                continue
            if ok_at(k, slot): pick = k; break
            if not allow_move: break
        if pick is None:
            nop = parse("s_nop 0"); out.append(nop); slot += 1; nops_new += 1
            continue
        i = seq[pick]; done[pick] = True; remaining -= 1
        slot_of[pick] = slot; order.append(pick); out.append(i); slot += 1
        for r in i.defs: last_writer[r] = pick
    # ---- second pass: fill the wait states that are left by SINKING an earlier instruction into them ----
    # The list scheduler above only pulls instructions UP.  A wait state in front of a consumer late in the block (the DPP reads of the step's
    # energy, one slot behind the v_cmp that follows its last addition) has nothing below it to pull up -- but an instruction further up whose
    # result is not needed until later (an exponent for the range guard, a compare whose mask is combined at the end) can be issued THERE
    # instead: the slot it leaves closes up, the wait state disappears.  Greedy, every candidate move is checked with the full re-check below
    # (dependencies in their old order, every hazard, every entry distance) before it is taken; LDS instructions, waits and branches never move.
    # (Sinking an instruction that merely READS or writes a register some LDS read loads to BEHIND an s_waitcnt is safe -- a wait only makes more
    # values valid; the rule "never across a wait" protects the other direction.  So the sink pass drops that one ordering: instruction -> later wait
    # through the LDSREGS pseudo register alone.  Real register dependencies -- e.g. on the ds_read that next loads the register -- stay.)
    def depends_sink(a, b):
        if b.kind == "wait" and a.kind != "wait":
            au = a.uses - {"LDSREGS"}
            return bool(a.defs & (b.uses | b.defs)) or bool(au & b.defs)
        return depends(a, b)
    preds_sink = [[a for a in preds[b] if depends_sink(seq[a], seq[b])] for b in range(n)]
    index_of = {id(i): k for k, i in enumerate(seq)}
    def valid(lst):
        pos_of, sl, lw = {}, 0, {}
        for x in lst:
            if x.kind != "nop": pos_of[index_of[id(x)]] = sl
            sl += 1
        for b in range(n):
            for a in preds_sink[b]:
                if pos_of[a] >= pos_of[b]: return False
        for x in lst:
            if x.kind == "nop": continue
            k = index_of[id(x)]
            if pos_of[k] < entry_min[k]: return False
            for r in x.uses:
                if r in lw:
                    need = hazard_need(seq[lw[r]], x, r, pk_wait)
                    if need and pos_of[k] - pos_of[lw[r]] - 1 < need: return False
            for r in x.defs: lw[r] = k
        return True
    sunk = 0
    if allow_move and nops_new:
        again = True
        while again:
            again = False
            for pos, x in enumerate(out):
                if x.kind != "nop": continue
                for cp in range(pos - 1, -1, -1):
                    c = out[cp]
                    if c.kind not in ("valu", "salu", "trans"): continue
                    cand = out[:cp] + out[cp + 1:pos] + [c] + out[pos + 1:]
                    if valid(cand):
                        out = cand; again = True; sunk += 1; nops_new -= 1; slot -= 1
                        break
                if again: break
        if sunk:   # positions changed: rebuild the bookkeeping the checks below use
            order, sl = [], 0
            for x in out:
                if x.kind != "nop":
                    k = index_of[id(x)]; order.append(k); slot_of[k] = sl
                sl += 1
    # values that LEAVE the block: the next block was scheduled by hipcc in the belief that every producer here sits at least as far from the
    # block's end as it did; with the wait states gone (or instructions pulled up in front of it) a producer may have come closer -- pad the
    # end until every producer keeps min(its old distance, the table's maximum)
    tail = 1 if term is not None else 0
    deficit = 0
    for k in range(n):
        i = seq[k]
        if i.kind in ("valu", "trans", "dpp") or (i.kind == "salu" and "m0" in i.defs):
            old_after = (old_slots - 1 - i.orig) + tail
            new_after = (slot - 1 - slot_of[k]) + tail
            deficit = max(deficit, min(old_after, HAZ_READ_MAX) - new_after)
    for _ in range(deficit):
        out.append(parse("s_nop 0")); slot += 1; nops_new += 1
    if term is not None:
        # the branch reads VCC / SCC a VALU / SALU instruction wrote: SALU reads are interlocked, no wait states; it stays last
        out.append(term)
    # ---- re-check: dependencies in their old order, hazards satisfied ----
    newpos = {k: p for p, k in enumerate(order)}
    for b in range(n):
        for a in (preds_sink[b] if sunk else preds[b]):
            assert newpos[a] < newpos[b], "dependency broken: %s -> %s" % (seq[a].text, seq[b].text)
    lw = {}
    for k in order:
        i = seq[k]
        assert slot_of[k] >= entry_min[k]
        for r in i.uses:
            if r in lw:
                need = hazard_need(seq[lw[r]], i, r, pk_wait)
                assert slot_of[k] - slot_of[lw[r]] - 1 >= need, "hazard: %s -> %s" % (seq[lw[r]].text, i.text)
        for r in i.defs: lw[r] = k
    assert sorted(id(x) for x in seq) == sorted(id(x) for x in out if x.kind != "nop" and x is not term)
    moved = sum(1 for p, k in enumerate(order) if k != p)
    return out, {"old_slots": old_slots + (1 if term else 0), "new_slots": slot + (1 if term else 0), "old_nops": sum(i.nop_states for i in body if i.kind == "nop"), "new_nops": nops_new, "moved": moved}


def process(lines, kernel_re, pk_wait, allow_move, report):
    out = []
    n = 0
    kre = re.compile(kernel_re)
    total = {"blocks": 0, "touched": 0, "old_nops": 0, "new_nops": 0, "moved": 0}
    per_kernel = {}
    while n < len(lines):
        l = lines[n]
        m = re.match(r"^([A-Za-z_][\w$.]*):", l)
        if not (m and kre.search(m.group(1)) and not m.group(1).startswith(".")):
            out.append(l); n += 1; continue
        name = m.group(1)
        end = n + 1
        while end < len(lines) and not lines[end].startswith(".Lfunc_end"): end += 1   # (a kernel may hold several s_endpgm)
        func = lines[n:end + 1]
        # registers loaded by any LDS read of this kernel
        ldsregs = set()
        for fl in func:
            t = fl.split(";")[0].strip()
            if t.startswith("ds_read"):
                ldsregs |= set(regs_of(split_ops(t.split(None, 1)[1])[0]))
        ks = per_kernel.setdefault(name, {"blocks": 0, "touched": 0, "old_nops": 0, "new_nops": 0, "moved": 0})
        block, block_lines = [], []
        def flush():
            nonlocal block, block_lines
            if not block:
                out.extend(block_lines); block, block_lines = [], []; return
            ks["blocks"] += 1
            has_nop = any(i.kind == "nop" for i in block)
            if has_nop and all(i.known for i in block) and not any(i.is_branch for i in block[:-1]):
                new, st = schedule_block(block, ldsregs, pk_wait, allow_move)
                ks["touched"] += 1; ks["old_nops"] += st["old_nops"]; ks["new_nops"] += st["new_nops"]; ks["moved"] += st["moved"]
                for i in new: out.append("\t" + i.text)
            else:
                if has_nop: ks["old_nops"] += sum(i.nop_states for i in block if i.kind == "nop"); ks["new_nops"] += sum(i.nop_states for i in block if i.kind == "nop")
                out.extend(block_lines)
            block, block_lines = [], []
        for fl in func:
            t = fl.split(";")[0].strip()
            is_label = bool(re.match(r"^[.\w$]+:", t))
            if not t:
                # comments / asm markers / blank lines inside a block are dropped with it; outside they are kept
                if block: block_lines.append(fl)
                else: out.append(fl)
                continue
            if t.startswith(".") and not is_label or is_label:
                flush(); out.append(fl); continue
            i = parse(t)
            block.append(i); block_lines.append(fl)
            if i.is_branch or not i.known or t.startswith(("s_endpgm", "s_setpc", "s_swappc", "s_barrier", "s_sleep", "s_sethalt", "s_trap")):
                flush()
        flush()
        n = end + 1
    for k, v in per_kernel.items():
        for f in total: total[f] += v[f]
    if report:
        with open(report, "w") as f:
            f.write("# asm_sched.py: wait states (s_nop states) of the straight-line blocks, hipcc -> after the pass; kernels matching /%s/, pk-wait %d, moves %s\n" % (kernel_re, pk_wait, "on" if allow_move else "off"))
            for k, v in sorted(per_kernel.items()):
                f.write("%-160s blocks %5d rescheduled %4d  s_nop states %4d -> %4d  instructions moved %5d\n" % (k[:160], v["blocks"], v["touched"], v["old_nops"], v["new_nops"], v["moved"]))
            f.write("total: %d kernels, blocks %d, rescheduled %d, s_nop states %d -> %d, instructions moved %d\n" % (len(per_kernel), total["blocks"], total["touched"], total["old_nops"], total["new_nops"], total["moved"]))
    return out, total, per_kernel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("inp"); ap.add_argument("out")
    ap.add_argument("--kernels", default=r"k_sweep2|k_sweep_t")
    ap.add_argument("--pk-wait", type=int, default=0)
    ap.add_argument("--no-move", action="store_true", help="only drop the wait states the table does not ask for; never reorder")
    ap.add_argument("--report")
    ap.add_argument("--dpp-old-wait", type=int, default=2, help="wait states behind the writer of a partial DPP move's OLD value (LLVM: 2, like the source; tests/micro/dpp_old_probe.hip: the hardware needs none)")
    a = ap.parse_args()
    global DPP_OLD_WAIT
    DPP_OLD_WAIT = a.dpp_old_wait
    lines = open(a.inp).read().split("\n")
    out, total, _ = process(lines, a.kernels, a.pk_wait, not a.no_move, a.report)
    open(a.out, "w").write("\n".join(out))
    print("asm_sched: %d blocks rescheduled, s_nop states %d -> %d, %d instructions moved" % (total["touched"], total["old_nops"], total["new_nops"], total["moved"]), file=sys.stderr)


if __name__ == "__main__":
    main()
