#!/usr/bin/env python3
"""Generates csrc/median_net.inl: the selection network of medianBlur(5) for TWO horizontally adjacent outputs.

medianBlur(flow, flow, 5) (CPU/PixFlow.hpp:325,338) is a pure selection -- the result is the element of rank 13 of the 25 values of
the window, whatever algorithm finds it -- so any correct selection network gives the reference's bits.  The kernels compute two
horizontally adjacent outputs per thread from 6 columns x 5 rows (columns 0-4 = output 0's window, columns 1-5 = output 1's).

Network (per channel; min/max exchanges only, no data-dependent control flow):
  1. sort each of the 6 columns: a 3-element sort is v_min3 / v_med3 / v_max3, and inserting d into a sorted run x0 <= .. <= xk is
     min(x0, d), med3(x0, d, x1), .., med3(x(k-1), d, xk), max(xk, d) -- 3 + 4 + 5 = 12 instructions per column instead of the 18 of
     the optimal 9-exchange network;
  2. merge columns 1+2 and 3+4 (Batcher odd-even merges of sorted sequences), then merge the two sorted tens: of the 20 values the
     two windows share only ranks 8..13 can be either window's median (an element with rank r in the shared set has rank r .. r+5
     in a window);
  3. per output: merge those six with the output's own fifth column (0 or 5) and take rank 6 of the 11.
     Rank 6 of sorted six s8..s13 and a sorted column c1..c5 needs no merge: the k-th smallest of two sorted lists is
     max over i + j = k - 1 of min(a[i+1], b[j+1]), here max(s8, min(s9,c5), min(s10,c4), min(s11,c3), min(s12,c2), min(s13,c1)):
     five v_min + two v_max3 + one v_max.
Everything the two outputs do not depend on is removed by a backward liveness pass (an exchange of which only the minimum or only
the maximum is used costs one instruction instead of two): 176 instructions per channel for two outputs, against ~345 for the
hand-written "forgetful selection" it replaces (kernels_level.hip, rounds 1-3).

(A 2 x 2 quad network -- a column's rows 1-4 sorted once for both output rows, 77.5 instead of 88 instructions per output and channel --
was generated, verified and measured in round 4: no gain at 36 live float2 inputs per thread, profiles/r04_median_tile_ab.txt.)
NaN: v_min / v_max return the other operand, v_med3 with a NaN operand returns the minimum of the other two -- a NaN acts as a
duplicate of a finite neighbour, the result is a finite member of the window (tests/test_gpu_hygiene.py defines exactly that).

The generated network is verified here before it is written: EXHAUSTIVELY with the 0-1 principle (a min/max network selects rank k
of every input iff it does so for every 0/1 input: all 2^30 of them, bit-parallel), and on random floats with ties.
Run:  python tools/gen_median_net.py   (rewrites csrc/median_net.inl; tests/test_median_net.py re-runs the verification on the CPU tier)
"""
import os
import sys

import numpy as np


class Net:
    def __init__(self, n_in):
        self.n_in = n_in
        self.ops = []          # (kind, a, b): value id len(ops)+n_in = min/max of values a, b
    def _op(self, kind, a, b):
        self.ops.append((kind, a, b))
        return self.n_in + len(self.ops) - 1
    def ce(self, a, b):
        return self._op("min", a, b), self._op("max", a, b)
    def _op3(self, kind, a, b, c):
        self.ops.append((kind, a, b, c))
        return self.n_in + len(self.ops) - 1
    def insert(self, run, d):   # run ascending
        out = [self._op("min", run[0], d)]
        for i in range(len(run) - 1):
            out.append(self._op3("med3", run[i], d, run[i + 1]))
        out.append(self._op("max", run[-1], d))
        return out
    def sort5(self, v):
        a, b, c, d, e = v
        run = [self._op3("min3", a, b, c), self._op3("med3", a, b, c), self._op3("max3", a, b, c)]
        return self.insert(self.insert(run, d), e)
    def select_mid(self, mid6, col):
        """rank 6 of the sorted six mid6 = s8..s13 and the sorted five col (see the module docstring)"""
        terms = [mid6[0]] + [self._op("min", mid6[1 + j], col[4 - j]) for j in range(5)]
        a = self._op3("max3", terms[0], terms[1], terms[2]); b = self._op3("max3", terms[3], terms[4], terms[5])
        return self._op("max", a, b)
    def pair_from_sorted(self, cols):
        """the two medians (columns 0-4, columns 1-5) from six SORTED columns of five"""
        s12 = self.merge(cols[1], cols[2])
        s34 = self.merge(cols[3], cols[4])
        shared = self.merge(s12, s34)
        mid6 = shared[7:13]                       # ranks 8..13 of the 20 shared values
        return self.select_mid(mid6, cols[0]), self.select_mid(mid6, cols[5])
    def merge(self, A, B):
        """Batcher's (m, n) odd-even merge of two ascending sequences of value ids (Knuth 5.3.4)."""
        if not A: return list(B)
        if not B: return list(A)
        if len(A) == 1 and len(B) == 1:
            return list(self.ce(A[0], B[0]))
        C = self.merge(A[0::2], B[0::2])
        D = self.merge(A[1::2], B[1::2])
        out = [C[0]]
        i = 0
        while i < len(D) and i + 1 < len(C):
            lo, hi = self.ce(D[i], C[i + 1])
            out += [lo, hi]
            i += 1
        return out + D[i:] + C[i + 1:]


def build():
    net = Net(30)
    cols = [net.sort5([c * 5 + r for r in range(5)]) for c in range(6)]
    return net, net.pair_from_sorted(cols)


def prune(net, outs):
    live = set(outs)
    keep = [False] * len(net.ops)
    for k in range(len(net.ops) - 1, -1, -1):
        vid = net.n_in + k
        if vid in live:
            keep[k] = True
            live.update(net.ops[k][1:])
    return keep


def evaluate(net, keep, outs, inputs, fmin, fmax):
    val = list(inputs) + [None] * len(net.ops)
    for k, op in enumerate(net.ops):
        if not keep[k]: continue
        kind, x = op[0], [val[i] for i in op[1:]]
        if kind == "min": r = fmin(x[0], x[1])
        elif kind == "max": r = fmax(x[0], x[1])
        elif kind == "min3": r = fmin(fmin(x[0], x[1]), x[2])
        elif kind == "max3": r = fmax(fmax(x[0], x[1]), x[2])
        else: r = fmax(fmin(x[0], x[1]), fmin(fmax(x[0], x[1]), x[2]))   # med3
        val[net.n_in + k] = r
    return [val[o] for o in outs]


def verify_01(net, keep, outs, log=print):
    """All 2^30 zero-one inputs, 64 per machine word: input i's bit pattern = bit i of the input index."""
    lowpat = [0xAAAAAAAAAAAAAAAA, 0xCCCCCCCCCCCCCCCC, 0xF0F0F0F0F0F0F0F0, 0xFF00FF00FF00FF00, 0xFFFF0000FFFF0000, 0xFFFFFFFF00000000]
    W = 1 << 18                                # words per chunk: index bits 6..23
    idx = np.arange(W, dtype=np.uint64)
    mid = [np.where((idx >> np.uint64(b)) & np.uint64(1), np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0)) for b in range(18)]
    ones = np.full(W, 0xFFFFFFFFFFFFFFFF, np.uint64); zeros = np.zeros(W, np.uint64)

    def majority13(wires):   # bit-sliced count of 25 one-bit inputs, then count >= 13
        cnt = [zeros.copy() for _ in range(5)]
        for w in wires:
            carry = w
            for b in range(5):
                t = cnt[b] & carry
                cnt[b] = cnt[b] ^ carry
                carry = t
        c0, c1, c2, c3, c4 = cnt
        # >= 13  <=>  c4 | (c3 & c2 & (c1 | c0))   (16..25 | 13,14,15)
        return c4 | (c3 & c2 & (c1 | c0))

    for chunk in range(64):
        ins = []
        for i in range(30):
            if i < 6: ins.append(np.full(W, lowpat[i], np.uint64))
            elif i < 24: ins.append(mid[i - 6])
            else: ins.append(ones if (chunk >> (i - 24)) & 1 else zeros)
        got = evaluate(net, keep, outs, ins, np.bitwise_and, np.bitwise_or)
        want0 = majority13(ins[0:25]); want1 = majority13(ins[5:30])
        if not (np.array_equal(got[0], want0) and np.array_equal(got[1], want1)):
            raise SystemExit("0-1 verification FAILED in chunk %d" % chunk)
    log("0-1 principle: all 2^30 inputs give the two medians")


def verify_random(net, keep, outs, n=200000, seed=7):
    rng = np.random.default_rng(seed)
    x = rng.integers(-6, 7, size=(30, n)).astype(np.float32) * np.float32(0.25)   # many ties
    x[:, : n // 2] = rng.standard_normal((30, n // 2)).astype(np.float32)
    got = evaluate(net, keep, outs, list(x), np.minimum, np.maximum)
    want0 = np.sort(x[0:25], axis=0)[12]; want1 = np.sort(x[5:30], axis=0)[12]
    assert np.array_equal(got[0], want0) and np.array_equal(got[1], want1), "random verification failed"


def emit(net, keep, outs):
    n_ops = sum(keep)
    L = ["// GENERATED by tools/gen_median_net.py -- do not edit.  medianBlur(5) for two horizontally adjacent outputs (CPU/PixFlow.hpp:325,338):",
         "// in[c * 5 + r] = column c (0..5), row r (0..4) of the 6 x 5 neighbourhood; o0 = median of columns 0-4, o1 = median of columns 1-5.",
         "// %d min / max / min3 / med3 / max3 instructions (sorted columns, odd-even merges, everything the two medians do not depend on pruned);" % n_ops,
         "// verified exhaustively with the 0-1 principle (2^30 inputs) by the generator and by tests/test_median_net.py.",
         "__device__ __forceinline__ void d_median_pair_net(const float* in, float& o0, float& o1) {"]
    name = lambda v: ("in[%d]" % v) if v < net.n_in else ("t%d" % (v - net.n_in))
    fn = {"min": "__builtin_fminf", "max": "__builtin_fmaxf", "min3": "d_min3", "max3": "d_max3", "med3": "__builtin_amdgcn_fmed3f"}
    for k, op in enumerate(net.ops):
        if keep[k]:
            L.append("  const float t%d = %s(%s);" % (k, fn[op[0]], ", ".join(name(v) for v in op[1:])))
    L.append("  o0 = %s; o1 = %s;" % (name(outs[0]), name(outs[1])))
    L.append("}")
    return "\n".join(L) + "\n", n_ops


def main(write=True, log=print):
    net, outs = build()
    keep = prune(net, outs)
    verify_random(net, keep, outs)
    verify_01(net, keep, outs, log)
    text, n_ops = emit(net, keep, outs)
    log("%d operations built, %d instructions live" % (len(net.ops), n_ops))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "csrc", "median_net.inl")
    if write:
        open(path, "w").write(text)
        log("wrote " + os.path.normpath(path))
    return text, os.path.normpath(path)


if __name__ == "__main__":
    main(write="--check" not in sys.argv)
