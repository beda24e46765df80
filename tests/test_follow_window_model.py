"""A CPU model of the sweeps' flow-following gather window (round 5; csrc/kernels_sweep2.hip loader of k_sweep2, csrc/kernels_sweep_t.inl):
the loader's bookkeeping restated in a few lines of Python and run against random offset sequences, to check the two claims the kernels'
comments make and the GPU tests can only confirm indirectly (a wrong texel shows up as a wrong flow somewhere):
  (1) RESIDENCY  -- whenever chunk j's window has been published, every texel a pixel of chunk j may sample under the step's test
                    (|sample - centre| <= test radius, centre = pixel + the chunk's offset) sits in the torus at its absolute slot;
  (2) NO CLOBBER -- what the loader writes for chunk j never replaces a texel that a chunk still in use (the loader runs at most `ahead` chunks
                    in front of its band) may sample.
Both sweep forms: latency (8 rows per band, 64 x 32 torus, window +-8, test 7, 3 chunks ahead) and throughput (32 rows, skewed 64 x 48 torus,
window +-6, test 5, 3 chunks ahead).  Offsets: a random walk with jumps (the loader limits it to one texel per chunk and axis) near the image
borders too (centres are cut back into the image)."""
import numpy as np
import pytest

CH = 8   # steps per chunk


class Model:
    def __init__(self, rows, rad, test, ring_rows, ahead, skew, LS, LB, vb, uLo=0, fill_back=False):
        self.R, self.rad, self.test, self.RV, self.ahead, self.skew = rows, rad, test, ring_rows, ahead, skew
        self.fill_back = fill_back   # throughput form: its window's lower edge can move back (see the loader); the latency form's cannot
        self.LS, self.LB, self.vb, self.uLo = LS, LB, vb, uLo
        self.tor = -np.ones((ring_rows, 64, 2), np.int64)      # what texel (u, v) each slot holds
        self.off = {}                                          # chunk -> applied (ou, ov)

    def slot(self, u, v):
        return (v % self.RV, (u + v) % 64 if self.skew else u % 64)

    def store(self, u, v, live):
        if not (0 <= u < self.LS and 0 <= v < self.LB):
            return
        r, c = self.slot(u, v)
        old = tuple(self.tor[r, c])
        assert old == (u, v) or old not in live, "chunk data clobbered: slot (%d, %d) held live texel %s, overwritten by (%d, %d)" % (r, c, old, u, v)
        self.tor[r, c] = (u, v)

    def clamp(self, j, ou, ov):
        plo = min(max(self.uLo + CH * j - (self.R - 1), 0), self.LS - 1); phi = min(max(self.uLo + CH * j + CH - 1, 0), self.LS - 1)
        vhi = min(self.vb + self.R - 1, self.LB - 1)
        return min(max(ou, -plo), self.LS - 1 - phi), min(max(ov, -self.vb), self.LB - 1 - vhi)

    def rect(self, j, ou, ov):
        """columns (or skew slots D = u + v) [lo, hi) and rows [r0, r1) of chunk j's window"""
        r0 = self.vb + ov - self.rad
        if self.skew:
            DB = self.uLo + self.vb + ou + ov
            return DB + CH * j - 2 * self.rad, DB + CH * j + CH + 2 * self.rad, r0, r0 + self.R + 2 * self.rad
        return self.uLo + CH * j - (self.R - 1) - self.rad + ou, self.uLo + CH * j + CH - 1 + self.rad + 1 + ou, r0, r0 + self.R + 2 * self.rad

    def needed(self, j):
        """texels a pixel of chunk j may sample: pixel (row r, step s) at (uLo + s - r, vb + r), centre + (ou, ov), sample within `test` of the centre,
        footprint floor .. floor + 1 (one more texel either side for a difference that rounds onto the bound: the kernels' windows have it)"""
        ou, ov = self.off[j]
        out = set()
        for r in range(self.R):
            v = self.vb + r
            if v >= self.LB:
                continue
            for s in range(CH * j, CH * j + CH):
                u = self.uLo + s - r
                if not (0 <= u < self.LS):
                    continue
                for du in range(-self.test - 1, self.test + 2):
                    for dv in range(-self.test - 1, self.test + 2):
                        uu, vv = u + ou + du, v + ov + dv
                        if 0 <= uu < self.LS and 0 <= vv < self.LB:
                            out.add((uu, vv))
        return out

    def run(self, targets):
        live_sets = {}
        po = None; front = None; pov = None
        for j, (tu, tv) in enumerate(targets):
            if po is not None:
                tu = min(max(tu, po[0] - 1), po[0] + 1); tv = min(max(tv, po[1] - 1), po[1] + 1)
            ou, ov = self.clamp(j, tu, tv)
            lo, hi, r0, r1 = self.rect(j, ou, ov)
            live = set().union(*[live_sets[k] for k in live_sets if k >= j - self.ahead]) if live_sets else set()
            cols = lambda a, b, rows: [((D - v, v) if self.skew else (D, v)) for D in range(a, b) for v in rows]
            if po is None:
                for (u, v) in cols(lo, hi, range(r0, r1)):
                    self.store(u, v, live)
                front = hi; have_lo = lo
            else:
                for (u, v) in cols(front, max(front, hi), range(r0, r1)):       # the new columns / skew slots of the chunk's rows
                    self.store(u, v, live)
                if ov != pov:                                                   # the row that entered, over what is already there
                    vnew = r1 - 1 if ov > pov else r0
                    for (u, v) in cols(lo, front, [vnew]):
                        self.store(u, v, live)
                    have_lo = max(have_lo, lo)
                if self.fill_back and lo < have_lo:                             # the lower edge moved back (far-border cut of the offset): fill in
                    for (u, v) in cols(lo, have_lo, range(r0, r1)):
                        self.store(u, v, live)
                    have_lo = lo
                front = max(front, hi)
            po = (tu, tv) if (ou, ov) == (tu, tv) else (ou, ov); pov = ov
            self.off[j] = (ou, ov)
            live_sets[j] = self.needed(j)
            # residency for every chunk the band may still be working on
            for k in range(max(0, j - self.ahead), j + 1):
                for (u, v) in live_sets[k]:
                    r, c = self.slot(u, v)
                    assert tuple(self.tor[r, c]) == (u, v), "chunk %d (after loading %d): texel (%d, %d) not resident (slot holds %s)" % (k, j, u, v, tuple(self.tor[r, c]))


def _walk(rng, n, jump_every, amp):
    o = np.zeros(2, np.int64); out = []
    for j in range(n):
        o += rng.integers(-2, 3, 2)
        if jump_every and j % jump_every == jump_every - 1:
            o += rng.integers(-amp, amp + 1, 2)
        out.append((int(o[0]), int(o[1])))
    return out


@pytest.mark.parametrize("seed", range(24))
def test_latency_form_window_model(seed):
    rng = np.random.default_rng(seed)
    LS, LB = 330, 70
    vb = int(rng.choice([0, 8, 32, 56, 64]))
    m = Model(rows=8, rad=8, test=7, ring_rows=32, ahead=3, skew=False, LS=LS, LB=LB, vb=vb)   # 32-step record ring: chunk j is loaded once the band is in chunk j - 3
    m.run(_walk(rng, (LS + 7 + 7) // CH + 1, jump_every=[0, 7, 11][seed % 3], amp=25))


@pytest.mark.parametrize("seed", range(24))
def test_throughput_form_window_model(seed):
    rng = np.random.default_rng(100 + seed)
    LS, LB = 300, 110
    vb = int(rng.choice([0, 32, 64, 96]))
    m = Model(rows=32, rad=6, test=5, ring_rows=48, ahead=3, skew=True, LS=LS, LB=LB, vb=vb, fill_back=True)
    m.run(_walk(rng, (LS + 31 + 7) // CH + 1, jump_every=[0, 9, 13][seed % 3], amp=25))
