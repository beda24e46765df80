"""ISA-level guards of the throughput-form sweep: its compute waves count the completion of their LDS-DMA record requests BY HAND
(`s_waitcnt vmcnt(tPre - 1)`, csrc/kernels_sweep_t.inl), which is only right while nothing else touches the wave's vector-memory counter in the
hot loop -- a register spill to scratch, or a global load the compiler hoisted into it, would show up as rare wrong records, not as a steady
failure.  Compiled here with the product's flags (no GPU needed):
  * no k_sweep_t instantiation uses scratch (private segment 0: spill code could land in any wave role);
  * the steady-state loop of every compute_band_t<TOP> holds exactly one global_load_lds_dwordx4 per step and no other vector-memory
    instruction between two of them (the out-of-window fallback's loads live in cold blocks outside the loop);
  * every k_sweep_t fits one workgroup per CU (LDS <= 160 KB)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "panorama-opticalflow_amd")
SRC = os.path.join(PKG, "csrc", "kernels_sweep2.hip")


def _make_vars():
    """HIPCC, the compile flags of the sweep TU and the scheduling pass's flags, read from the Makefile (one source of truth)"""
    mk = open(os.path.join(PKG, "Makefile")).read()
    var = lambda name: re.search(r"^%s\s*\??=\s*(.*)$" % re.escape(name), mk, re.M).group(1).strip()
    hipcc = os.environ.get("HIPCC") or var("HIPCC")
    arch = os.environ.get("ARCH") or var("ARCH")
    flags = var("FLAGS").replace("$(ARCH)", arch).split() + var("FLAGS_kernels_sweep2").split()
    return hipcc, flags, var("SCHED_FLAGS").split(), var("LLVM_BIN")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc, flags, _, _ = _make_vars()
    if shutil.which(hipcc) is None:
        pytest.skip("no hipcc on this host (%s): the ISA guards need the ROCm compiler" % hipcc)
    out = str(tmp_path_factory.mktemp("isa") / "sweep2.s")
    subprocess.check_call([hipcc] + flags + ["-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.fixture(scope="module")
def asm_product(asm, tmp_path_factory):
    """what ships: hipcc's assembly after tools/asm_sched.py with the Makefile's flags"""
    _, _, sched_flags, _ = _make_vars()
    d = tmp_path_factory.mktemp("isa_sched")
    src, dst = str(d / "in.s"), str(d / "out.s")
    open(src, "w").write(asm)
    subprocess.check_call([sys.executable, os.path.join(PKG, "tools", "asm_sched.py"), src, dst] + sched_flags, stderr=subprocess.DEVNULL)
    return open(dst).read()


def _demangle(names):
    _, _, _, llvm_bin = _make_vars()
    filt = os.path.join(llvm_bin, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = shutil.which("c++filt")
    if filt is None:
        return {n: n for n in names}
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def _kernels(asm, prefix):
    meta = {}
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.vgpr_count:\s+(\d+)", asm, re.S):
        if prefix in m.group(2):
            meta[m.group(2)] = {"lds": int(m.group(1)), "scratch": int(m.group(3)), "vgprs": int(m.group(4))}
    return meta


def _latency_forward_kernel(asm):
    """mangled name of k_sweep2<SwGeom<4, 1>, TR = false, FWD = true, SPARSE = false, MODE = 0>, found by its demangled name"""
    names = sorted(_kernels(asm, "k_sweep2"))
    dem = _demangle(names)
    hits = [n for n in names if re.search(r"k_sweep2<.*SwGeom<4, 1>, false, true, false, 0>", dem[n])]
    if not hits:   # no demangler: the mangled fragment
        hits = [n for n in names if "SwGeomILi4ELi1EEELb0ELb1ELb0ELi0" in n]
    assert len(hits) == 1, (hits, list(dem.values())[:4])
    return hits[0]


def test_throughput_form_has_no_scratch_and_fits_the_cu(asm):
    meta = _kernels(asm, "k_sweep_t")
    assert len(meta) == 4, sorted(meta)            # <TR, FWD> x 2 x 2
    for name, m in meta.items():
        assert m["scratch"] == 0, "%s spills %d bytes to scratch" % (name, m["scratch"])
        assert m["lds"] <= 160 * 1024, (name, m["lds"])
        assert m["vgprs"] <= 168, (name, m["vgprs"])   # 11 waves per workgroup: three per SIMD


def test_throughput_form_hot_loop_has_one_record_request_per_step(asm):
    names = sorted(_kernels(asm, "k_sweep_t"))
    for name in names:
        body = asm[asm.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")]
        lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
        dma = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
        # three compute_band_t<TOP> instances, each: tPre = 6 requests in front of the loop + 8 (the unrolled chunk)
        assert len(dma) == 3 * (6 + 8), (name, len(dma))
        steps = 0
        for inst in range(3):
            loop = dma[inst * 14 + 6:inst * 14 + 14]      # the unrolled chunk's eight requests (the six before them are the prologue's)
            assert all(b - a < 12 for a, b in zip(dma[inst * 14:inst * 14 + 6], dma[inst * 14 + 1:inst * 14 + 6])), (name, "prologue")
            for a, b in zip(loop, loop[1:]):              # seven whole steps between them
                assert 150 < b - a < 320, (name, b - a)   # (a step is ~230 instructions)
                steps += 1
                other = [l for l in lines[a + 1:b] if re.match(r"(global_|scratch_|buffer_|flat_)", l)]
                assert not other, "%s: vector-memory instruction inside the hot loop: %s" % (name, other[:3])
                waits = [l for l in lines[a + 1:b] if l.startswith("s_waitcnt") and "vmcnt" in l]
                assert waits and all("vmcnt(5)" in w for w in waits), (name, waits)   # the hand-counted wait, and no compiler-inserted vmcnt(0)
        assert steps == 3 * 7, (name, steps)


def _steps_of(asm_text, name, lo, hi):
    body = asm_text[asm_text.index("\n" + name + ":"):]
    body = body[:body.index("\n.Lfunc_end")]      # (the kernel has several s_endpgm: one per wave role)
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    # a step starts at its `row_newbcast:0` DPP move (round 6: ONE 64-bit move per step)
    starts = [i for i, l in enumerate(lines) if l.startswith("v_mov_b64_dpp") and "row_newbcast:0" in l]
    return lines, [(a, b) for a, b in zip(starts, starts[1:]) if lo <= b - a <= hi]   # consecutive steps of the unrolled chunks


def test_latency_form_step_keeps_its_issue_slots_down(asm, asm_product):
    """A lone wave issues one instruction (or s_nop) per ~4 cycles whatever it is (tests/micro/slot_model.py), so the step costs its issue SLOTS.
    Round 6 took it from 115 to 103: 64-bit DPP moves for the proposals, the across moves folded into v_cndmask_b32_dpp / v_sub_f32_dpp, the torus
    addressed by image coordinates, the gradient division's guard on the energies, and the scheduling pass filling every wait state.  A compiler
    or source change that brings slots back costs ~1 % of every sweep each without failing any parity test: caught here, on the assembly that
    ships (hipcc -> tools/asm_sched.py with the Makefile's flags).  (Dense, not transposed, forward.)"""
    name = _latency_forward_kernel(asm)
    lines, steps = _steps_of(asm_product, name, 90, 125)
    assert len(steps) >= 3 * 7, len(steps)                                          # three compute_band<TOP> instances x seven whole steps
    slots = [b - a for a, b in steps]
    assert max(slots) <= 112 and sum(slots) / len(slots) <= 106.0, slots           # (101-104 today; the chunk's first step carries the chunk's bookkeeping)
    nops = [sum(int(l.split()[1], 0) + 1 for l in lines[a:b] if l.startswith("s_nop")) for a, b in steps]
    assert max(nops) <= 2 and sum(nops) / len(nops) <= 1.0, nops                    # hipcc leaves 7-9 per step; the pass none in most
    fused = 0
    for a, b in steps:
        seg = lines[a:b]
        assert sum(1 for l in seg if l.startswith("v_mov_b64_dpp")) == 3, "the proposals' three 64-bit DPP moves"
        fused_here = (sum(1 for l in seg if l.startswith("v_cndmask_b32_dpp")), sum(1 for l in seg if l.startswith("v_sub_f32_dpp")))
        assert fused_here in ((2, 2), (0, 0)), fused_here       # (a sweep's first band -- no cross-lane neighbour for row 0 -- keeps the plain selection)
        fused += fused_here == (2, 2)
        # the two sum-of-squares blocks: one packed add per block that reads its operand with swapped halves
        assert sum(1 for l in seg if l.startswith("v_pk_add_f32") and "op_sel:[0,1] op_sel_hi:[1,0]" in l) == 2
    assert fused >= 2 * 7, fused


def test_dpp_reads_inside_asm_blocks_keep_their_wait_states(asm, asm_product):
    """The compiler does not see a DPP read inside an asm statement: the blocks carry their own s_nop where the source is young (the finite
    differences read the energy's last addition), and the scheduling pass re-derives the two wait states.  In EVERY sweep kernel of the product
    assembly a v_sub_f32_dpp / v_cndmask_b32_dpp must stand at least two slots behind the last VALU write of its DPP source."""
    bad = []
    for name in sorted(_kernels(asm, "k_sweep2")):
        body = asm_product[asm_product.index("\n" + name + ":"):]
        body = body[:body.index("\n.Lfunc_end")]
        lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", "."))]
        for i, l in enumerate(lines):
            if not l.startswith(("v_sub_f32_dpp", "v_cndmask_b32_dpp")):
                continue
            src = l.split(None, 1)[1].split(",")[1].strip()          # src0 = the DPP operand
            dist = 0
            for k in range(i - 1, max(i - 4, -1), -1):
                p = lines[k]
                if p.endswith(":"):
                    break
                if p.startswith("v_") and re.match(r"\S+\s+%s\b" % re.escape(src), p) or (p.startswith("v_") and re.match(r"\S+\s+v\[(\d+):(\d+)\]", p) and
                                                                                          int(re.match(r"\S+\s+v\[(\d+):(\d+)\]", p).group(1)) <= int(src[1:]) <= int(re.match(r"\S+\s+v\[(\d+):(\d+)\]", p).group(2))):
                    if dist < 2:
                        bad.append((name[-40:], p, l))
                    break
                dist += 2 if p.startswith("s_nop 1") else 1
    assert not bad, bad[:3]
