#!/bin/bash
# Build container (no GPU): per kernel of the product's translation units, how many memory instructions are FLAT (address space not
# inferred: counted in lgkmcnt too, every wait on one is vmcnt(0) lgkmcnt(0)), how many waits are vmcnt(0) against counted vmcnt(N),
# and scratch traffic.  Found with it in round 4: PF_BOFF's uintptr_t round trip made every access of every batched kernel FLAT,
# and a generic lambda put k_gauss15_fused's prefetch registers into scratch.
cd "$(dirname "$0")/../../panorama-opticalflow_amd"
mkdir -p /tmp/isa
for f in kernels_pre kernels_level kernels_misc kernels_sweep2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $( [ $f = kernels_sweep2 ] && echo "-mllvm -amdgpu-sched-strategy=max-ilp" ) -S --cuda-device-only -o /tmp/isa/lint_$f.s csrc/$f.hip 2>/dev/null
done
python3 - <<'PY'
import re
for f in ('kernels_pre', 'kernels_level', 'kernels_misc', 'kernels_sweep2'):
    lines = open('/tmp/isa/lint_%s.s' % f).read().split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_ZN2pf\w+:', l)]
    for (i, name), (j, _) in zip(starts, starts[1:] + [(len(lines), '')]):
        body = lines[i:j]
        ld = sum(1 for l in body if re.match(r'^\s+(global|flat)_load', l))
        fl = sum(1 for l in body if re.match(r'^\s+flat_', l))
        w0 = sum(1 for l in body if re.search(r's_waitcnt.*vmcnt\(0\)', l))
        wn = sum(1 for l in body if re.search(r's_waitcnt.*vmcnt\([1-9]', l))
        sc = sum(1 for l in body if 'scratch_' in l)
        print("%-8s %-60s loads %4d flat %4d vmcnt(0) %4d vmcnt(N) %4d scratch %3d" % (f[8:], name[6:66], ld, fl, w0, wn, sc))
PY
