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
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "panorama-opticalflow_amd", "csrc", "kernels_sweep2.hip")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("isa") / "sweep2.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                           "-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm, prefix):
    meta = {}
    for m in re.finditer(r"\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.vgpr_count:\s+(\d+)", asm, re.S):
        if prefix in m.group(2):
            meta[m.group(2)] = {"lds": int(m.group(1)), "scratch": int(m.group(3)), "vgprs": int(m.group(4))}
    return meta


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


def test_latency_form_step_keeps_its_wait_states_down(asm):
    """Round 5: for a lone wave an s_nop costs 4 cycles -- as much as an instruction (tests/micro/nop_cost.hip) -- and the step is issue-bound.
    hipcc put 8 into every step; the window-address block and the two sum-of-squares blocks (csrc/exact_forms.hpp) took three out.  A compiler or
    source change that brings them back costs ~2 % of every sweep without failing any parity test: caught here.  (Dense, not transposed, forward.)"""
    names = [n for n in _kernels(asm, "k_sweep2") if "SwGeomILi4ELi1EEELb0ELb1ELb0ELi0" in n]
    assert len(names) == 1, names
    body = asm[asm.index("\n" + names[0] + ":"):]
    body = body[:body.index("\n.Lfunc_end")]      # (the kernel has several s_endpgm: one per wave role)
    lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    # a step starts at its first `row_newbcast:0` DPP move (two per step: x and y of the proposal)
    starts = [i for i, l in enumerate(lines) if "row_newbcast:0" in l and (i == 0 or "row_newbcast:0" not in lines[i - 1])]
    steps = [(a, b) for a, b in zip(starts, starts[1:]) if 100 <= b - a <= 140]     # consecutive steps of the unrolled chunks (a step is ~119 instructions)
    assert len(steps) >= 3 * 7, len(steps)                                          # three compute_band<TOP> instances x seven whole steps
    nops = [sum(1 for l in lines[a:b] if l.startswith("s_nop")) for a, b in steps]
    assert max(nops) <= 7 and sum(nops) / len(nops) <= 6.0, nops     # (5-7 today, 5.7 on average over the three loop instances; the profiled instance had 8 before the blocks)
    # and the blocks are there: one packed add per block that reads its operand with swapped halves
    swaps = [sum(1 for l in lines[a:b] if l.startswith("v_pk_add_f32") and "op_sel:[0,1] op_sel_hi:[1,0]" in l) for a, b in steps]
    assert min(swaps) == 2 and max(swaps) == 2, swaps
