#!/usr/bin/env python
"""VALU floor of the throughput mode, re-derived from MEASURED issue rates (round-4 review, weak #3 / next #1b).  No GPU needed.

profiles/r05_valu_issue_raw.txt (tests/micro/valu_issue.hip on an MI355X) gives the cycles one SIMD needs per wave64 instruction with 8 waves
resident, per instruction class.  This script compiles the kernel files to gfx950 assembly with the product's flags, takes every hot kernel's
STATIC vector-instruction histogram (its whole text: the hot loops dominate it), prices it with those classes, and multiplies the resulting
cycles per vector instruction with the DYNAMIC per-pair instruction counts of a batch (profiles/r04_batch_instruction_counts.txt, rocprofv3
--pmc SQ_INSTS_VALU): the time 1024 SIMDs need to issue a dense pair's vector instructions, i.e. the batch's VALU floor."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "panorama-opticalflow_amd", "csrc")
# cycles per wave64 instruction of a saturated SIMD (N = 8 waves), profiles/r05_valu_issue_raw.txt
FULL, HALF, QUARTER = 2.25, 4.1, 8.1
FULL_OPS = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
            "v_cndmask_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32", "v_accvgpr", "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32")
QUARTER_OPS = ("v_rsq_f32", "v_sqrt_f32", "v_rcp_f32", "v_exp_f32", "v_log_f32", "v_permlane32_swap", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_div_fmas_f64", "v_div_fixup_f64")


def cost(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.endswith("_dpp") or "dpp" in op: return HALF
    if base.startswith(QUARTER_OPS): return QUARTER
    if base.startswith(FULL_OPS): return FULL
    return HALF   # packed fp32, v_max / v_min / v_med3 / v_min3, v_cmp, conversions, v_fract, v_frexp, v_lshl_add, v_mad_u32_u24, ... (measured half rate)


def kernels(asm):
    out = {}
    for m in re.finditer(r"\n(_Z\S+):\s*;\s*@", asm):
        name = m.group(1)
        body = asm[m.end():]
        body = body[:body.index("s_endpgm")] if "s_endpgm" in body else body
        h = collections.Counter()
        for l in body.splitlines():
            t = l.strip().split()
            if t and t[0].startswith("v_"): h[t[0]] += 1
        out[name] = h
    return out


FAMILIES = [("k_sweep_t", "k_sweep_t<true>"), ("k_median5_tiled", "k_median5_tiled"), ("8k_sweep2INS_12_GLOBAL__N_16SwGeomILi4ELi1EEELb0ELb1ELb0ELi0", "k_sweep2<pf::>"), ("k_gauss15_fused", "k_gauss15_fused"),
            ("k_sweep_prepILi32", "k_sweep_prep<32>"), ("k_sweep_prepILi8", "k_sweep_prep<8>"), ("k_upsample_cubic_tiled", "k_upsample_cubic_tiled"), ("k_final_flow", "k_final_flow"),
            ("k_gradients_all", "k_gradients_all"), ("k_pyr_down4", "k_pyr_down4"), ("k_blend_batch", "k_blend_batch"), ("k_downscale_gray", "k_downscale_gray")]


def main():
    dyn = {}
    for l in open(os.path.join(ROOT, "profiles", "r04_batch_instruction_counts.txt")):
        t = l.split()
        if len(t) >= 3 and t[0].startswith("k_"):
            try: dyn[t[0]] = float(t[1])
            except ValueError: pass
    hists = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("kernels_sweep2", "kernels_level", "kernels_pre", "kernels_misc"):
            out = os.path.join(tmp, f + ".s")
            extra = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if f == "kernels_sweep2" else []
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"] + extra + ["-S", "--cuda-device-only", "-o", out, os.path.join(CS, f + ".hip")],
                                  stderr=subprocess.DEVNULL)
            hists.update(kernels(open(out).read()))
    print("# cycles per wave64 vector instruction of a saturated SIMD (8 waves): full rate %.2f (v_fma / v_mul / v_add / v_mov / v_and / v_add_u32 / v_cndmask), half rate %.1f (packed fp32, DPP," % (FULL, HALF))
    print("# v_max / v_med3 / v_min3, v_cmp, conversions, v_fract, v_frexp, v_lshl_add, v_mad_u32_u24), quarter rate %.1f (v_rsq / v_sqrt / v_rcp, v_permlane32_swap, fp64)" % QUARTER)
    print("%-28s %10s %8s %8s %8s %12s %14s" % ("kernel family", "static v_*", "full %", "half %", "quart %", "cycles/instr", "M instr / pair"))
    tot_c = tot_n = 0.0
    for key, fam in FAMILIES:
        h = collections.Counter()
        for name, hh in hists.items():
            if key in name: h.update(hh)
        n = sum(h.values())
        if not n: continue
        cyc = sum(cost(op) * c for op, c in h.items())
        fr = lambda v: 100.0 * sum(c for op, c in h.items() if cost(op) == v) / n
        d = dyn.get(fam, 0.0)
        print("%-28s %10d %8.1f %8.1f %8.1f %12.2f %14.1f" % (fam, n, fr(FULL), fr(HALF), fr(QUARTER), cyc / n, d))
        tot_c += cyc / n * d; tot_n += d
    allv = dyn.get("total", tot_n)
    avg = tot_c / tot_n
    print("# listed kernels: %.0f M of the pair's %.0f M vector instructions; weighted %.2f cycles per instruction" % (tot_n, allv, avg))
    for ghz in (2.4, 2.2):
        ms = allv * 1e6 * avg / (1024 * ghz * 1e9) * 1e3
        print("# VALU floor per dense pair = %.0f M x %.2f cycles / (1024 SIMDs x %.1f GHz) = %.2f ms" % (allv, avg, ghz, ms))


if __name__ == "__main__":
    main()
