#!/usr/bin/env python3
"""CPU prototype of the plan costed in notes/r3_experiments.md: evaluate a strictly sequential fp32 sum of NON-NEGATIVE
terms (the reference's splat of one lattice vertex, permutohedral_cpu.h:653-661) exactly, but with the bulk of the
additions done in parallel.

    s = +0;  for p in terms:  s = fl(s + p)                                  (what must be reproduced bit for bit)

While s stays inside one binade [2^k, 2^(k+1)) every partial sum is a multiple of u = 2^(k-23), and
fl(s + p) = s + (p rounded to a multiple of u, ties to the multiple that makes the SUM even).  So a block of terms that
lies inside one binade can be added from a stand-in start value -- 2^k, or 2^k + u for the other parity of s/u, which
matters only until the block's first tie, after which both chains agree -- by somebody else, in parallel, and the block's
exact increment added to the true partial sum later: S + A is exact (multiples of u below 2^(k+1)).  A parity-dependent
block (A0 != A1, always |A1 - A0| = u) needs no branch in the sequential pass either: the final parity is that of A0 in both
cases, so adding the half-way value (A0 + A1) / 2 lets round-to-nearest-EVEN pick the right neighbour when that parity is
even, and (A0 + A1) / 2 - u followed by + u when it is odd.

Which blocks lie inside one binade is decided from an approximate prefix sum (any summation order of non-negative floats is
within n 2^-24 relative of the true sum, and so is the sequential result): a block is `certain` when its start and end,
widened by that bound, have the same exponent.  The other blocks (the first few -- the sum crosses a binade every few terms
while it is small -- and the ~log2(sum / first term) later crossings) are added term by term.

This file is not part of the product and nothing imports it; `python scripts/exact_sum_prototype.py` runs the
self-check (random SLAM-like rows and adversarial rows against the plain loop)."""
import sys

import numpy as np

F32 = np.float32
UNIT = 8                                                  # terms per block (one ring unit of fused_loop.h's chain_rows)


def sequential(terms):
    s = F32(0.0)
    for p in terms:
        s = F32(s + p)
    return s


def exponent(x):
    return int((np.float32(x).view(np.uint32) >> 23) & 0xff)


def segmented(terms, stats=None):
    """Same result as sequential(terms); returns None when the row is not eligible (a negative or non-finite term)."""
    t = np.asarray(terms, F32)
    n = len(t)
    if n == 0:
        return F32(0.0)
    if not np.all(np.isfinite(t)) or np.any(np.signbit(t) & (t != 0)):
        return None
    nb = (n + UNIT - 1) // UNIT
    pad = np.zeros(nb * UNIT, F32)
    pad[:n] = t
    blocks = pad.reshape(nb, UNIT)
    # ---- parallel part 1: approximate prefix at block boundaries (float64 here; any order of fp32 adds would do) -------
    bsum = blocks.astype(np.float64).sum(axis=1)
    hi_pref = np.cumsum(bsum)
    lo_pref = hi_pref - bsum
    delta = max(n, 64) * 2.0 ** -22                       # generous: four times the n 2^-24 bound
    kind = np.zeros(nb, np.int8)                          # 0 raw, 1 pure, 2 parity-dependent
    A0 = np.zeros(nb, F32)
    A1 = np.zeros(nb, F32)
    kk = np.zeros(nb, np.int32)
    for j in range(nb):
        lo, hi = F32(lo_pref[j] * (1.0 - delta)), F32(hi_pref[j] * (1.0 + delta))
        k = exponent(lo)
        if not (lo > 0 and k == exponent(hi) and 0 < k < 254):
            continue
        # ---- parallel part 2: the block from both stand-in starts -------------------------------------------------
        c0 = np.uint32(k << 23).view(F32)                 # 2^(k-127)
        c1 = np.uint32((k << 23) + 1).view(F32)           # ... + u
        z0, z1 = c0, c1
        for p in blocks[j]:
            z0 = F32(z0 + p)
            z1 = F32(z1 + p)
        if exponent(z1) != k:                              # cannot happen when the bound holds; be safe
            continue
        a0 = int(z0.view(np.uint32)) - int(c0.view(np.uint32))          # increments in units of u
        a1 = int(z1.view(np.uint32)) - int(c1.view(np.uint32))
        u = np.ldexp(1.0, k - 127 - 23)
        A0[j], A1[j] = F32(a0 * u), F32(a1 * u)
        kk[j] = k
        kind[j] = 1 if a0 == a1 else 2
        assert abs(a0 - a1) <= 1
    # ---- sequential part: raw blocks term by term, one addition per run of pure blocks, one or two per tie block -----
    s = F32(0.0)
    adds = 0
    j = 0
    while j < nb:
        if kind[j] == 0:
            for p in blocks[j]:
                s = F32(s + p)
            adds += UNIT
            j += 1
        elif kind[j] == 1:
            run = 0.0                                      # exact: integers times u, below 2^24 u
            k = kk[j]
            while j < nb and kind[j] == 1 and kk[j] == k:
                run += float(A0[j])
                j += 1
            assert exponent(s) == k
            s = F32(s + F32(run))
            adds += 1
        else:
            k = kk[j]
            assert exponent(s) == k
            u = np.ldexp(1.0, int(k) - 127 - 23)
            mid = F32((float(A0[j]) + float(A1[j])) / 2.0)            # min(A0, A1) + u / 2: 24 bits at most
            if int(round(float(A0[j]) / u)) % 2 == 0:
                s = F32(s + mid)
                adds += 1
            else:
                s = F32(F32(s + F32(float(mid) - u)) + F32(u))
                adds += 2
            j += 1
    if stats is not None:
        stats.append((n, adds, int((kind == 0).sum()), int((kind == 2).sum())))
    return s


def slam_like_row(rng, n):
    bary = rng.random(n).astype(F32) * F32(0.8) + F32(0.05)
    q = np.where(rng.random(n) < 0.2, rng.random(n) * 1e-3, 1.0 - rng.random(n) * 1e-3).astype(F32)
    return (bary * q).astype(F32)


def adversarial_row(rng, n):
    kind = rng.integers(0, 7)
    if kind == 0:                                          # EVERY addition a tie: the sum sits in [1, 2) (u = 2^-23, or in [2, 4) once
        t = (rng.integers(0, 4096, n) * 2.0 ** -23 + 2.0 ** -24).astype(F32)     # it has crossed), every term ends in half an ulp
        t[0] = F32(rng.choice([1.0, 1.5, 1.9990234375]))
        return t
    if kind == 6:                                          # ties now and then, on sums of any size
        return (rng.integers(0, 64, n) * 2.0 ** -20 + (rng.random(n) < 0.3) * 2.0 ** -21 + (rng.random(n) < 0.2) * 2.0 ** -28).astype(F32)
    if kind == 1:                                          # powers of two, sums landing exactly on binade boundaries
        return (2.0 ** rng.integers(-12, 3, n)).astype(F32)
    if kind == 2:                                          # zeros, tiny and subnormal terms between ordinary ones
        t = slam_like_row(rng, n)
        t[rng.random(n) < 0.3] = 0.0
        t[rng.random(n) < 0.1] = F32(1e-42)
        t[rng.random(n) < 0.1] = F32(3e-39)
        return t
    if kind == 3:                                          # huge dynamic range
        return (rng.random(n) * 10.0 ** rng.uniform(-30, 8, n)).astype(F32)
    if kind == 4:                                          # every term the same
        return np.full(n, F32(rng.random() + 0.01), F32)
    t = slam_like_row(rng, n)                              # a large term in the middle: a jump over several binades
    t[n // 2] = F32(10.0 ** rng.uniform(0, 6))
    return t


def self_check(rows=4000, seed=1):
    rng = np.random.default_rng(seed)
    stats = []
    for i in range(rows):
        n = int(rng.integers(1, 700))
        t = slam_like_row(rng, n) if i % 2 == 0 else adversarial_row(rng, n)
        want, got = sequential(t), segmented(t, stats if i % 2 == 0 else None)
        assert got is not None and want.view(np.uint32) == got.view(np.uint32), (i, n, want, got)
    assert segmented(np.array([1.0, -0.5], F32)) is None and segmented(np.array([np.inf], F32)) is None
    s = np.array([x for x in stats if x[0] >= 300])
    print("%d rows bit-identical to the plain loop; SLAM-like rows of >= 300 terms: %.0f sequential additions instead of %.0f "
          "(%.1f raw blocks, %.1f tie blocks per row)" % (rows, s[:, 1].mean(), s[:, 0].mean(), s[:, 2].mean(), s[:, 3].mean()))


if __name__ == "__main__":
    self_check(int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
