#!/usr/bin/env python
"""Round 6: what does the issue stage of gfx950 charge ONE wave alone on its SIMD for an instruction, depending on what stands in front of it?
(The latency-form sweep lives in that regime: DESIGN.md 3.2.)  Generates tests/micro/slot_model.hip -- one kernel per pattern, every body a
pure asm loop with explicit registers, timed with s_memtime on a lone wave per CU -- builds it and (on the GPU box) runs it:
    python tests/micro/slot_model.py gen      # writes slot_model.hip + builds tests/micro/slot_model   (build container, no GPU)
    tests/micro/slot_model                    # on the GPU box: cycles per body / per instruction for every pattern
The numbers feed the cost model of panorama-opticalflow_amd/tools/asm_sched.py (the post-pass scheduler of the sweep's step)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PATTERNS = []   # (name, body lines (one repetition unit), repetitions, instructions counted per unit)


def pat(name, unit, reps, n=None):
    PATTERNS.append((name, unit, reps, n if n is not None else len([l for l in unit if not l.startswith("s_nop")])))


def fill(i, kind="add"):
    r = 8 + (i % 8)
    if kind == "add":
        return "v_add_f32 v%d, v%d, v1" % (r, r)
    if kind == "pk":
        r = 16 + 2 * (i % 8)
        return "v_pk_add_f32 v[%d:%d], v[%d:%d], v[32:33]" % (r, r + 1, r, r + 1)
    if kind == "salu":
        return "s_add_u32 s%d, s%d, 1" % (24 + i % 4, 24 + i % 4)
    raise ValueError(kind)


# ---- dependent chains with k independent fillers between the links ----
for k in range(0, 5):
    pat("v_add chain, %d independent v_add between links" % k, ["v_add_f32 v0, v0, v1"] + [fill(j) for j in range(k)], 64 // (k + 1) * 2)
pat("v_add chain, dependent through src1", ["v_add_f32 v0, v1, v0"], 96)
pat("v_mul -> v_add alternating chain", ["v_mul_f32 v0, v0, v1", "v_add_f32 v0, v0, v1"], 48)
pat("v_fma chain (src2 carries)", ["v_fma_f32 v0, v1, v1, v0"], 96)
pat("v_fma chain (src0 carries)", ["v_fma_f32 v0, v0, v1, v1"], 96)
for k in range(0, 4):
    pat("v_pk_add chain, %d independent v_pk_add between links" % k, ["v_pk_add_f32 v[2:3], v[2:3], v[32:33]"] + [fill(j, "pk") for j in range(k)], 64 // (k + 1) * 2)
pat("v_pk_add chain, 1 independent v_add between links", ["v_pk_add_f32 v[2:3], v[2:3], v[32:33]", fill(0)], 48)
pat("v_pk_fma chain, 1 independent v_add between links", ["v_pk_fma_f32 v[2:3], v[2:3], v[32:33], v[32:33]", fill(0)], 48)
pat("v_add chain, 1 s_add_u32 between links", ["v_add_f32 v0, v0, v1", fill(0, "salu")], 48)
pat("v_add chain, 2 s_add_u32 between links", ["v_add_f32 v0, v0, v1", fill(0, "salu"), fill(1, "salu")], 32)
pat("v_add chain, s_nop 0 between links", ["v_add_f32 v0, v0, v1", "s_nop 0"], 48, 1)
pat("independent v_add + s_add_u32 alternating", [fill(0), fill(0, "salu"), fill(1), fill(1, "salu"), fill(2), fill(2, "salu"), fill(3), fill(3, "salu")], 12)
pat("independent v_add x8", [fill(j) for j in range(8)], 12)
pat("independent v_pk_add x8", [fill(j, "pk") for j in range(8)], 12)
pat("independent s_add_u32 x4", [fill(j, "salu") for j in range(4)], 24)
# ---- consumers of other kinds ----
pat("v_add -> v_pk_add (reads it) -> v_add chain", ["v_add_f32 v2, v2, v1", "v_pk_add_f32 v[2:3], v[2:3], v[32:33]"], 48)
pat("v_rsq chain", ["v_rsq_f32 v0, v0"], 64)
pat("v_rsq -> s_nop 0 -> v_add (reads it), chained", ["v_rsq_f32 v0, v0", "s_nop 0", "v_add_f32 v0, v0, v1"], 32, 2)
pat("v_rsq -> v_add filler -> v_add (reads it), chained", ["v_rsq_f32 v0, v0", fill(0), "v_add_f32 v0, v0, v1"], 32)
# ---- DPP reads of a fresh VALU result: wait states by s_nop or by independent work ----
DPP = "v_mov_b32_dpp v0, v0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
pat("v_add ; s_nop 1 ; v_mov_dpp (reads it), chained", ["v_add_f32 v0, v0, v1", "s_nop 1", DPP], 32, 2)
pat("v_add ; 2 independent v_add ; v_mov_dpp (reads it), chained", ["v_add_f32 v0, v0, v1", fill(0), fill(1), DPP], 24)
pat("v_add ; 3 independent v_add ; v_mov_dpp (reads it), chained", ["v_add_f32 v0, v0, v1", fill(0), fill(1), fill(2), DPP], 24)
pat("v_mov_dpp ; v_add (reads it) chained, s_nop 1 in front of the dpp", ["s_nop 1", DPP, "v_add_f32 v0, v0, v1"], 32, 2)
pat("v_mov_dpp chain (s_nop 1 each)", ["s_nop 1", DPP], 48, 1)
pat("v_mov_dpp x4 independent sources", ["v_mov_b32_dpp v%d, v%d row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" % (8 + j, 40 + j) for j in range(4)], 24)
pat("v_mov_b64_dpp row_newbcast x4 independent", ["v_mov_b64_dpp v[%d:%d], v[%d:%d] row_newbcast:0 row_mask:0xf bank_mask:0x9" % (16 + 2 * j, 17 + 2 * j, 40 + 2 * j, 41 + 2 * j) for j in range(4)], 24)
pat("v_mov_b32_dpp row_newbcast x4 independent", ["v_mov_b32_dpp v%d, v%d row_newbcast:0 row_mask:0xf bank_mask:0x9" % (16 + j, 40 + j) for j in range(4)], 24)
pat("v_mov_b64_dpp newbcast ; v_pk_add (reads it) ; s_nop 1, chained", ["s_nop 1", "v_mov_b64_dpp v[2:3], v[2:3] row_newbcast:0 row_mask:0xf bank_mask:0xf", "v_pk_add_f32 v[2:3], v[2:3], v[32:33]"], 32, 2)
pat("2 x v_mov_b32_dpp newbcast ; v_pk_add (reads them) ; s_nop 1, chained", ["s_nop 1", "v_mov_b32_dpp v2, v2 row_newbcast:0 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp v3, v3 row_newbcast:0 row_mask:0xf bank_mask:0xf", "v_pk_add_f32 v[2:3], v[2:3], v[32:33]"], 24, 3)
# ---- compare -> select ----
pat("v_cmp (sgpr pair) ; v_cndmask (reads it) chained, s_nop 1 between", ["v_cmp_lt_f32 s[30:31], v0, v1", "s_nop 1", "v_cndmask_b32 v0, v0, v1, s[30:31]"], 32, 2)
pat("v_cmp (sgpr pair) ; 2 independent v_add ; v_cndmask chained", ["v_cmp_lt_f32 s[30:31], v0, v1", fill(0), fill(1), "v_cndmask_b32 v0, v0, v1, s[30:31]"], 24)
pat("v_cmp (vcc) ; v_cndmask (vcc) chained", ["v_cmp_lt_f32 vcc, v0, v1", "v_cndmask_b32 v0, v0, v1, vcc"], 48)
# ---- LDS ----
pat("ds_read_b64 -> wait -> v_add on the address, chained (latency)", ["ds_read_b64 v[4:5], v40", "s_waitcnt lgkmcnt(0)", "v_add_f32 v0, v0, v4"], 32, 2)
pat("ds_read2_b64 x2 -> wait -> use, chained (latency)", ["ds_read2_b64 v[4:7], v40 offset1:1", "ds_read2_b64 v[44:47], v40 offset0:67 offset1:68", "s_waitcnt lgkmcnt(0)", "v_add_f32 v0, v0, v44"], 24, 3)
for k in (8, 12, 16, 20, 24, 28):
    pat("ds_read2_b64 x2 ; %d independent v_add ; wait ; use" % k, ["ds_read2_b64 v[4:7], v40 offset1:1", "ds_read2_b64 v[44:47], v40 offset0:67 offset1:68"] + [fill(j) for j in range(k)] + ["s_waitcnt lgkmcnt(0)", "v_add_f32 v0, v0, v44"], 8, k + 3)
pat("ds_read_b128 x3 + ds_read_b32 + ds_read_b64 among 8 independent v_add", ["ds_read_b128 v[4:7], v40", fill(0), "ds_read_b128 v[44:47], v40 offset:16", fill(1), "ds_read_b128 v[48:51], v40 offset:32", fill(2), "ds_read_b32 v52, v40", fill(3),
     "ds_read_b64 v[54:55], v40", fill(4), fill(5), fill(6), fill(7), "s_waitcnt lgkmcnt(0)"], 8, 13)
pat("ds_write_b64 + ds_write_b32 among 8 independent v_add", ["ds_write_b64 v40, v[0:1]", fill(0), "ds_write_b32 v40, v1 offset:8", fill(1), fill(2), fill(3), fill(4), fill(5), fill(6), fill(7)], 10)
# ---- branch not taken, waitcnt with nothing pending ----
pat("independent v_add x4 ; s_cbranch_vccnz (not taken)", [fill(0), fill(1), fill(2), fill(3), "s_cbranch_vccnz 2f"], 20)
pat("independent v_add x4 ; s_waitcnt lgkmcnt(0) (nothing pending)", [fill(0), fill(1), fill(2), fill(3), "s_waitcnt lgkmcnt(0)"], 20)

# ---- incremental cost of ONE instruction of a kind inside a stream of 24 independent v_add (results waited for at the end of the unit) ----
BASE24 = [fill(j) for j in range(24)]
def incr(name, ins, tail=None):
    pat("24 v_add + " + name, BASE24[:12] + ins + BASE24[12:] + (tail or []), 4, 24)
pat("24 v_add (baseline of the incremental costs)", BASE24, 4, 24)
incr("s_waitcnt lgkmcnt(0) at the end only", [], ["s_waitcnt lgkmcnt(0)"])
for w in ("ds_write_b32 v40, v1", "ds_write_b64 v40, v[0:1]", "ds_write_b96 v40, v[0:2]", "ds_write_b128 v40, v[0:3]"):
    incr(w.split()[0], [w], ["s_waitcnt lgkmcnt(0)"])
incr("ds_write_b64 + ds_write_b32 (the step's publish)", ["ds_write_b64 v40, v[0:1]", "ds_write_b32 v40, v1 offset:1024"], ["s_waitcnt lgkmcnt(0)"])
for r in ("ds_read_b32 v44, v40", "ds_read_b64 v[44:45], v40", "ds_read_b96 v[44:46], v40", "ds_read_b128 v[44:47], v40", "ds_read2_b64 v[44:47], v40 offset1:1", "ds_read2_b32 v[44:45], v40 offset1:1"):
    incr(r.split()[0], [r], ["s_waitcnt lgkmcnt(0)"])
incr("ds_read_b128 x3 (48-byte lane stride, as the records)", ["ds_read_b128 v[44:47], v41", "ds_read_b128 v[48:51], v41 offset:16", "ds_read_b128 v[52:55], v41 offset:32"], ["s_waitcnt lgkmcnt(0)"])
incr("ds_read_b128 x2 (32-byte lane stride)", ["ds_read_b128 v[44:47], v42", "ds_read_b128 v[48:51], v42 offset:16"], ["s_waitcnt lgkmcnt(0)"])
incr("v_rsq_f32", ["v_rsq_f32 v44, v45"])
incr("v_rsq_f32 x2", ["v_rsq_f32 v44, v45", "v_rsq_f32 v46, v47"])
incr("s_cbranch_vccnz (not taken)", ["s_cbranch_vccnz 2f"])
incr("s_cbranch_scc1 (not taken)", ["s_cmp_eq_u32 s24, -1", "s_cbranch_scc1 2f"])
incr("s_nop 0", ["s_nop 0"])
incr("s_nop 1", ["s_nop 1"])
incr("v_readfirstlane_b32", ["v_readfirstlane_b32 s26, v45"])
incr("v_cmp_lt_f32 vcc", ["v_cmp_lt_f32 vcc, v44, v45"])
incr("v_cmp_lt_f32 sgpr pair", ["v_cmp_lt_f32 s[30:31], v44, v45"])
incr("v_cndmask_b32 (sgpr pair mask)", ["v_cndmask_b32 v44, v45, v46, s[30:31]"])
incr("v_pk_fma_f32", ["v_pk_fma_f32 v[44:45], v[46:47], v[48:49], v[50:51]"])
incr("v_pk_add_f32", ["v_pk_add_f32 v[44:45], v[46:47], v[48:49]"])
incr("v_fma_f32", ["v_fma_f32 v44, v45, v46, v47"])
incr("v_med3_f32", ["v_med3_f32 v44, v45, v46, v47"])
incr("v_min3_i32", ["v_min3_i32 v44, v45, v46, v47"])
incr("v_mad_u32_u24", ["v_mad_u32_u24 v44, v45, v46, v47"])
incr("v_mov_b32_dpp row_newbcast", ["v_mov_b32_dpp v44, v45 row_newbcast:0 row_mask:0xf bank_mask:0x9"])
incr("v_mov_b64_dpp row_newbcast", ["v_mov_b64_dpp v[44:45], v[46:47] row_newbcast:0 row_mask:0xf bank_mask:0x9"])
incr("v_mov_b32_dpp row_bcast:15", ["v_mov_b32_dpp v44, v45 row_bcast:15 row_mask:0xe bank_mask:0x2"])
incr("s_or_b64 + s_and_b64", ["s_or_b64 s[30:31], s[30:31], vcc", "s_and_b64 vcc, s[30:31], vcc"])
incr("v_cmp_eq_u64", ["v_cmp_eq_u64 vcc, v[44:45], v[46:47]"])
incr("v_frexp_exp_i32_f32", ["v_frexp_exp_i32_f32 v44, v45"])

# ---- issue cost of LDS instructions with their latency hidden: the instruction(s) first, then 44 independent v_add, then the wait ----
BASE44 = [fill(j) for j in range(44)]
def incr_lds(name, ins):
    pat("LDS-hidden: " + name + " ; 44 v_add ; wait", ins + BASE44 + ["s_waitcnt lgkmcnt(0)"], 3, 44)
incr_lds("(nothing: baseline)", [])
for w in ("ds_write_b32 v40, v1", "ds_write_b64 v40, v[0:1]", "ds_write_b96 v41, v[0:2]", "ds_write_b128 v41, v[0:3]", "ds_write_b64 v41, v[0:1]", "ds_write_b128 v43, v[0:3]"):
    incr_lds(w.split(",")[0], [w])
incr_lds("ds_write_b64 + ds_write_b32 (the step's publish)", ["ds_write_b64 v40, v[0:1]", "ds_write_b32 v40, v1 offset:1024"])
for r in ("ds_read_b32 v44, v40", "ds_read_b64 v[44:45], v40", "ds_read_b96 v[44:46], v41", "ds_read_b128 v[44:47], v41", "ds_read_b128 v[44:47], v43", "ds_read2_b64 v[44:47], v40 offset1:1", "ds_read2_b64 v[44:47], v40 offset0:67 offset1:68"):
    incr_lds(r.split(",")[0] + (" " + r.split()[-2] + r.split()[-1] if "offset" in r else ""), [r])
incr_lds("ds_read_b128 x3 (48-byte lane stride, as the records)", ["ds_read_b128 v[44:47], v41", "ds_read_b128 v[48:51], v41 offset:16", "ds_read_b128 v[52:55], v41 offset:32"])
incr_lds("ds_read_b128 x2 (32-byte lane stride)", ["ds_read_b128 v[44:47], v42", "ds_read_b128 v[48:51], v42 offset:16"])
incr_lds("ds_read2_b64 x2 (the texels)", ["ds_read2_b64 v[44:47], v40 offset1:1", "ds_read2_b64 v[48:51], v40 offset0:67 offset1:68"])
incr_lds("ds_read_b32 + ds_read_b64 (counter + top value)", ["ds_read_b32 v44, v40", "ds_read_b64 v[46:47], v40 offset:2048"])
incr_lds("all nine LDS instructions of a step", ["ds_read2_b64 v[44:47], v40 offset1:1", "ds_read2_b64 v[48:51], v40 offset0:67 offset1:68", "ds_read_b128 v[52:55], v41", "ds_read_b128 v[56:59], v41 offset:16", "ds_read_b128 v[60:63], v41 offset:32",
          "ds_read_b32 v4, v40", "ds_read_b64 v[6:7], v40 offset:2048", "ds_write_b64 v40, v[0:1] offset:4096", "ds_write_b32 v40, v1 offset:5120"])


def gen():
    out = ["// generated by tests/micro/slot_model.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <vector>", ""]
    clob = ", ".join('"v%d"' % i for i in range(64)) + ', "s20", "s21", "s24", "s25", "s26", "s27", "s30", "s31", "vcc", "scc", "memory"'
    for n, (name, unit, reps, cnt) in enumerate(PATTERNS):
        body = "\\n\\t".join(unit * reps)
        out.append("__global__ void k%d(unsigned long long* rec, float* o, int iters) {" % n)
        out.append("  __shared__ float lds[8192]; for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = 1.0f + i * 1e-6f; __syncthreads();")
        out.append("  unsigned long long t0, t1; float r; const unsigned la = (unsigned)(size_t)lds + threadIdx.x * 8, lb = (unsigned)(size_t)lds + threadIdx.x * 48, lc = (unsigned)(size_t)lds + threadIdx.x * 32, ld = (unsigned)(size_t)lds + threadIdx.x * 16;")
        init = "\\n\\t".join(["v_mov_b32 v%d, 1.0" % i for i in range(64) if i not in (40, 41, 42, 43)] + ["v_mov_b32 v1, 0x3f800001", "v_mov_b32 v32, 0x3f800001", "v_mov_b32 v33, 0x3f800001", "v_mov_b32 v40, %3", "v_mov_b32 v41, %5", "v_mov_b32 v42, %6", "v_mov_b32 v43, %7",
                              "s_mov_b32 s24, 0", "s_mov_b32 s25, 0", "s_mov_b32 s26, 0", "s_mov_b32 s27, 0", "s_mov_b64 vcc, 0", "s_mov_b32 s20, %4"])
        out.append('  asm volatile("%s\\n\\ts_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_memtime %%0\\n\\ts_waitcnt lgkmcnt(0)\\n1:\\n\\t%s\\n\\ts_sub_u32 s20, s20, 1\\n\\ts_cmp_lg_u32 s20, 0\\n\\ts_cbranch_scc1 1b\\n2:\\n\\ts_waitcnt lgkmcnt(0)\\n\\ts_memtime %%1\\n\\ts_waitcnt lgkmcnt(0)\\n\\tv_add_f32 %%2, v0, v2"'
                   % (init, body))
        out.append('               : "=&s"(t0), "=&s"(t1), "=v"(r) : "v"(la), "s"(iters), "v"(lb), "v"(lc), "v"(ld) : %s);' % clob)
        out.append("  o[blockIdx.x * 64 + threadIdx.x] = r + lds[threadIdx.x]; if (threadIdx.x == 0) rec[blockIdx.x] = t1 - t0;")
        out.append("}")
    out.append("struct T { const char* name; void (*k)(unsigned long long*, float*, int); int per_body, nunit; };")
    out.append("static const T tests[] = {")
    for n, (name, unit, reps, cnt) in enumerate(PATTERNS):
        out.append('  {"%s", k%d, %d, %d},' % (name, n, cnt * reps, reps))
    out.append("};")
    out.append("""int main() {
  const int grid = 8, iters = 2000;
  float* o; unsigned long long* rec; hipMalloc(&o, grid * 64 * 4); hipMalloc(&rec, grid * 8);
  printf("# lone wave (one 64-thread workgroup per CU on %d CUs), cycles by s_memtime; per unit = one repetition of the pattern, per instr = counted instructions (s_nop / s_waitcnt not counted)\\n", grid);
  for (const T& t : tests) {
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(t.k, dim3(grid), dim3(64), 0, 0, rec, o, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), rec, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += double(v); c /= grid;
    printf("%-76s %8.2f cycles per unit  %6.2f per instr\\n", t.name, c / iters / t.nunit, c / iters / t.per_body);
  }
  return 0;
}""")
    src = os.path.join(HERE, "slot_model.hip")
    open(src, "w").write("\n".join(out) + "\n")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-o", os.path.join(HERE, "slot_model"), src])


if __name__ == "__main__":
    gen()
