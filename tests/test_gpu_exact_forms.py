"""The sweep's cheap exact forms (csrc/exact_forms.hpp: sqrt via v_rsq_f32 / v_sqrt_f32 + FMA corrections, division by a
constant with one FMA refinement) checked EXHAUSTIVELY on the device against the correctly rounded operations: every float
the sweep's range guard admits (zero, 2^-96 .. 2^100), ~2e9..4e9 inputs per form.  The hardware approximations behind them
cannot be emulated on a CPU, so this is the proof that the fast path computes the reference's sqrtf and '/' bit for bit
(CPU/PixFlow.hpp:258-277 errorFunction, :322-341 the gradient step)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN0 = os.path.join(HERE, "..", "panorama-opticalflow_amd", "tools", "exact_forms_check")


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["", "_safe"])
def test_exact_forms_exhaustive(which):
    """"" = the product's forms (asm-block packed chains), "_safe" = the compiler-scheduled forms of -DPF_SAFE_PK (the lab build's)"""
    BIN = BIN0 + which
    assert os.path.exists(BIN), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "mismatches 0"
    sq = [ln for ln in lines if ln.startswith("sqrt_core")][0]
    assert int(sq.split("tested")[1].split()[0]) > 1_600_000_000   # 196 binades x 2^23 mantissas + zero
    dv = [ln for ln in lines if ln.startswith("div by kGradEpsilon")][0]
    assert int(dv.split("tested")[1].split()[0]) > 3_200_000_000   # both signs


PROBE = os.path.join(HERE, "..", "panorama-opticalflow_amd", "tools", "pk_hazard_probe")


@pytest.mark.gpu
def test_packed_fp32_chains_need_no_wait_states():
    """csrc/exact_forms.hpp issues dependent v_pk_*_f32 instructions back to back (inside asm blocks), without the s_nop the compiler
    puts between them (its dst_sel hazard test misreads op_sel_hi of VOP3P as DST_OP_SEL).  The probe runs the same dependent chains with
    and without wait states -- one wave alone up to 16 waves per SIMD -- and must get identical bits."""
    assert os.path.exists(PROBE), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([PROBE], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical with and without s_nop" in r.stdout
    assert "differing results 0" in r.stdout


DPP_PROBE = os.path.join(HERE, "..", "panorama-opticalflow_amd", "tools", "dpp_old_probe")


@pytest.mark.gpu
def test_partial_dpp_writes_need_no_wait_states_behind_the_old_value():
    """The product's sweep TU is scheduled by tools/asm_sched.py with --dpp-old-wait 0: a DPP move whose bank / row masks leave lanes unwritten
    may directly follow the instruction that wrote its destination (LLVM counts the tied old operand like the DPP source: 2 wait states).  The
    probe runs the step's own sequence -- VALU producer, 64-bit row_newbcast moves, 32-bit row_bcast:15 moves -- with and without wait states,
    one wave alone up to 16 waves per SIMD, and must get identical bits."""
    assert os.path.exists(DPP_PROBE), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([DPP_PROBE], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical with and without wait states" in r.stdout
    assert r.stdout.strip().splitlines()[-1] == "mismatches 0"
