import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
import orc
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth"); orc.build()
ctx = pf.Context(0)
cols, rows = 480, 320
top, imgs = synth.make_stitch_set(cols, rows, 77, 5)
top = top.numpy(); imgs = [im.numpy() for im in imgs[:3]]
R = top; Rg = top
for i, L in enumerate(imgs):
    mp, ovl, ovr, blend, md = orc.stitch_prepare(L, R, True)
    f0, f1 = orc.flow_bidir(ovl, ovr, 20)
    merged = orc.combine_novel_views(ovl, ovr, f0, f1, blend)
    Rn = orc.stitch_gather(L, R, merged, mp)
    # GPU with the SAME inputs (oracle R)
    gmp, govl, govr, gblend, gmd = ctx.stitch_prepare(L, R)
    gm, g0, g1 = ctx.novel_view(govl, govr, 20, gblend)
    gR = ctx.stitch_gather(L, R, gm, gmp)
    print("step", i + 1, "map eq", np.array_equal(gmp, mp), "blend eq", np.array_equal(gblend, blend), "flow eq", np.array_equal(g0, f0), np.array_equal(g1, f1),
          "merged diff>0 %.5f >1 %.5f" % ((gm != merged).mean(), (np.abs(gm.astype(int) - merged.astype(int)) > 1).mean()),
          "final diff>0 %.5f" % (gR != Rn).mean(), "overlap frac %.3f" % (mp == 150).mean())
    # GPU chained on its own outputs
    cmp_, covl, covr, cblend, _ = ctx.stitch_prepare(L, Rg)
    cm, _, _ = ctx.novel_view(covl, covr, 20, cblend)
    Rg = ctx.stitch_gather(L, Rg, cm, cmp_)
    d = np.abs(Rg.astype(int) - Rn.astype(int))
    print("   chained: diff>0 %.5f  >1 %.5f  max %d" % ((d > 0).mean(), (d > 1).mean(), d.max()))
    R = Rn
