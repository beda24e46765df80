"""Stand-alone kernel times of the stencil stages at level-0 size of the strip (1000 x 2000), nothing else on the GPU: HIP-event time
per launch from the built-in profiler."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi")
c = pf.Context(0)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 2000)
r = np.random.default_rng(0)
flow = r.standard_normal((h, w, 2)).astype(np.float32)
a = np.ones((h, w), np.float32)
c.profile_enable(1)
for name, fn in (("median5", lambda: c.stage_median5(flow)), ("gauss15", lambda: c.stage_gauss(flow, 15, 8.0)), ("diffusion", lambda: c.stage_diffusion(a, a, flow)),
                 ("upsample", lambda: c.stage_upsample_cubic(flow[: int(h * 0.9), : int(w * 0.9)].copy(), w, h, 1.0 / 0.9))):
    fn(); c.profile_reset()
    for _ in range(5): fn()
    print(name, {k: "%.1f us x %d" % (1000 * ms / n, n) for k, (ms, n) in c.profile().items() if n})
    c.profile_reset()
