#!/usr/bin/env python3
"""CPU simulation for VERDICT r3 item 2: how many distinct 128-byte lines does a wavefront's blur gather touch on the C5 lattice,
by vertex numbering?  (numpy restatement of the lattice for LOCALITY statistics only -- nothing here is product or checker.)

  python scripts/sim_vertex_order.py [N]

Orders compared:
  ref      the reference's numbering: first occurrence over points in the caller's order
  zpoints  shipped locality mode: points along a 16-bit Z-order curve of their lattice cells, vertices by first occurrence
  zkeys    VERDICT's proposal: after that, vertices renumbered by the Z-order code of their OWN key (cell = key / (d+1))
  zkeys21  the same with 21 code bits (3.5 per dimension) instead of 16
  hilbert-ish gray   Gray-coded Z-order of the keys (neighbouring codes differ in one bit)
Metric: k_blur2's four gather instructions per wavefront (64 lanes x 2 vertices): distinct 128-byte lines among the lanes' addresses
(16 vertices per line, absent neighbours all read the zero vertex's line), averaged over wavefronts and the 7 axes; and the share of
present out-of-simplex neighbours within 4096 ids.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wl = importlib.import_module("lc-crf-slam_amd.workloads")


def lattice(f):
    N, d = f.shape
    D1 = d + 1
    inv_std = np.float32(np.sqrt(2.0 / 3.0) * D1)
    scale = np.array([np.float32(1.0 / np.sqrt(float((i + 2) * (i + 1))) * inv_std) for i in range(d)], np.float32)
    el = np.zeros((N, D1), np.float32)
    sm = np.zeros(N, np.float32)
    for j in range(d, 0, -1):
        cf = f[:, j - 1] * scale[j - 1]
        el[:, j] = sm - np.float32(j) * cf
        sm = sm + cf
    el[:, 0] = sm
    v = np.rint(el * np.float32(1.0 / D1)).astype(np.float32)
    rem0 = v * np.float32(D1)
    s = v.sum(1)
    diff = el - rem0
    rank = np.zeros((N, D1), np.float32)
    for i in range(d):
        for j in range(i + 1, D1):
            c = (diff[:, i] < diff[:, j]).astype(np.float32)
            rank[:, i] += c
            rank[:, j] += 1 - c
    rank += s[:, None]
    add = np.where(rank < 0, D1, 0) - np.where(rank >= D1, D1, 0)
    rank += add
    rem0 += add
    # keys of the d+1 corners: rem0 + canonical[rem][rank]
    keys = np.empty((N, D1, d), np.int32)
    for rem in range(D1):
        canon = np.where(rank[:, :d] <= d - rem, rem, rem - D1)
        keys[:, rem, :] = (rem0[:, :d] + canon).astype(np.int32)
    return keys, np.rint(el[:, :d] * np.float32(1.0 / D1)).astype(np.int32)


def pack(k):
    """(M, d) int32 keys -> (M,) void rows for np.unique / searchsorted"""
    k16 = np.ascontiguousarray((k + 32768).astype(np.uint16))
    return k16.view(np.dtype((np.void, k16.shape[1] * 2))).ravel()


def zcode(cells, bits):
    d = cells.shape[1]
    lo = cells.min(0)
    span = cells.max(0) - lo + 1
    rem = span.astype(np.int64).copy()
    nb = np.zeros(d, int)
    for _ in range(bits):
        b = int(np.argmax(rem))
        nb[b] += 1
        rem[b] = (rem[b] + 1) >> 1
    q = ((cells - lo).astype(np.int64) << nb) // span
    code = np.zeros(len(cells), np.int64)
    pos = 0
    for level in range(bits):
        for j in range(d):
            if nb[j] > level and pos < bits:
                code |= ((q[:, j] >> level) & 1) << pos
                pos += 1
    return code


def number_by_first_occurrence(ukey_of_entry, order_points, D1):
    """vertex id = rank of first occurrence over entries (point-major) with the points taken in `order_points`"""
    e = (order_points[:, None] * D1 + np.arange(D1)[None, :]).ravel()
    u = ukey_of_entry[e]
    _, first = np.unique(u, return_index=True)
    ids = np.empty(u.max() + 1, np.int64)
    ids[u[np.sort(first)]] = np.arange(len(first))
    return ids                                  # unique-key index -> vertex id


DUMP = None


def evaluate(name, ids, nbr_u, V):
    """nbr_u[j][u] = unique-key index of n1 / n2 along axis j (or -1)"""
    inv = np.empty(V, np.int64)
    inv[ids] = np.arange(V)                     # vertex id -> unique-key index
    if DUMP and name in DUMP.split(","):        # neighbour tables [axis][V][2] (int32, ids in THIS numbering) for scripts/ubench/blurorder.hip
        tab = np.empty((len(nbr_u), V, 2), np.int32)
        for j in range(len(nbr_u)):
            for side in range(2):
                nu = nbr_u[j][side][inv]
                tab[j, :, side] = np.where(nu >= 0, ids[np.maximum(nu, 0)], -1)
        os.makedirs(os.path.join(ROOT, "scripts", "ubench", "data"), exist_ok=True)
        fn = os.path.join(ROOT, "scripts", "ubench", "data", "nbr_%s.bin" % name)
        with open(fn, "wb") as fh:
            fh.write(np.array([V, len(nbr_u)], np.int32).tobytes())
            fh.write(tab.tobytes())
        print("   wrote", fn)
    lines_tot = 0.0
    near = far = 0
    nwaves = (V + 127) // 128
    pad = nwaves * 128 - V
    for j in range(len(nbr_u)):
        for side in range(2):
            nu = nbr_u[j][side][inv]            # per vertex id: neighbour's unique index
            nid = np.where(nu >= 0, ids[np.maximum(nu, 0)], -1)
            present = nid >= 0
            dist = np.abs(nid - np.arange(V))
            oos = present & (dist > 6)
            near += int((oos & (dist <= 4096)).sum())
            far += int((oos & (dist > 4096)).sum())
            line = np.where(present, (nid + 2) // 16, -1)         # the zero vertex sits in front of vertex 0
            line = np.concatenate([line, np.full(pad, -1)]).reshape(nwaves, 64, 2)
            for sub in range(2):                # the instruction of vertex 2t / of vertex 2t + 1
                l = np.sort(line[:, :, sub], axis=1)
                lines_tot += (np.diff(l, axis=1) != 0).sum() + nwaves
    n_instr = len(nbr_u) * 2 * 2 * nwaves
    print("%-10s lines per gather instruction %.2f   out-of-simplex neighbours within 4096 ids: %.1f %%" % (name, lines_tot / n_instr, 100.0 * near / max(near + far, 1)))


def main():
    global DUMP
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    DUMP = sys.argv[2] if len(sys.argv) > 2 else None
    f = wl.bilateral_problem(N, 1)["kernels"][0][0]
    d = f.shape[1]
    D1 = d + 1
    keys, cells = lattice(f)
    flat = keys.reshape(-1, d)
    uk, inv = np.unique(pack(flat), return_inverse=True)
    V = len(uk)
    ukeys = np.zeros((V, d), np.int32)
    ukeys[inv] = flat
    print("N %d  V %d" % (N, V))
    nbr_u = []
    for j in range(D1):
        pair = []
        for sgn in (-1, +1):                    # n1 = key - 1 (coordinate j: + d), n2 = key + 1 (coordinate j: - d)
            q = ukeys + sgn
            if j < d:
                q[:, j] -= sgn * D1
            pq = pack(q)
            pos = np.searchsorted(uk, pq)
            pos = np.minimum(pos, V - 1)
            pair.append(np.where(uk[pos] == pq, pos, -1))
        nbr_u.append(pair)
    ident = np.arange(N)
    evaluate("ref", number_by_first_occurrence(inv, ident, D1), nbr_u, V)
    code = zcode(cells, 16)
    zp = np.lexsort((ident, code))
    ids_zp = number_by_first_occurrence(inv, zp, D1)
    evaluate("zpoints", ids_zp, nbr_u, V)
    vcells = np.floor_divide(ukeys, D1)
    for bits, nm in ((16, "zkeys"), (21, "zkeys21"), (30, "zkeys30")):
        c = zcode(vcells, bits)
        order = np.lexsort((ids_zp, c))         # ties: the shipped order
        ids = np.empty(V, np.int64)
        ids[order] = np.arange(V)
        evaluate(nm, ids, nbr_u, V)
    # the lattice in the basis of its blur directions: c_j = (x_d - x_j) / (d + 1) with x_d = -sum_j x_j.  There the vertices are
    # plain integer grid points, axis j < d is the unit step along coordinate j and axis d the main diagonal.
    xd = -ukeys.sum(1)
    cc = (xd[:, None] - ukeys) // D1
    assert np.all((xd[:, None] - ukeys) % D1 == 0)
    for bits, nm in ((21, "zc21"), (30, "zc30"), (36, "zc36")):
        c = zcode(cc, bits)
        order = np.lexsort((ids_zp, c))
        ids = np.empty(V, np.int64)
        ids[order] = np.arange(V)
        evaluate(nm, ids, nbr_u, V)
    # row-major over c with the axes taken in the order 0..d-1 (axis 0 fastest): axis-0 neighbours are consecutive ids
    lo = cc.min(0)
    order = np.lexsort(tuple(cc[:, j] - lo[j] for j in range(d)))
    ids = np.empty(V, np.int64)
    ids[order] = np.arange(V)
    evaluate("rowmajor", ids, nbr_u, V)
    def order_ids(keys_minor_to_major):
        o = np.lexsort(tuple(keys_minor_to_major))
        i = np.empty(V, np.int64)
        i[o] = np.arange(V)
        return i
    cs = cc - lo
    span = cs.max(0) + 1
    print("c spans", span.tolist())
    # points in row-major order of their remainder-0 vertex's grid coordinates, vertices by first occurrence (no vertex sort at all)
    r0u = inv.reshape(N, D1)[:, 0]
    pc = cs[r0u]
    prm = np.lexsort(tuple([ident] + [pc[:, j] for j in range(d)]))
    ids_prm = number_by_first_occurrence(inv, prm, D1)
    evaluate("rmpoints", ids_prm, nbr_u, V)
    # class-major first occurrence: a vertex only ever appears as the corner of ONE remainder class, so number class 0's vertices
    # first (in the points' order), then class 1's, ...: d+1 runs, each nearly sorted by its own code when the points are sorted by
    # the code of their remainder-0 corner.  No vertex sort: only the scan order of the entries changes.
    def class_major(porder):
        e = (porder[None, :] * D1 + np.arange(D1)[:, None]).ravel()     # (rem, point) order
        u = inv[e]
        _, first = np.unique(u, return_index=True)
        ids = np.empty(V, np.int64)
        ids[u[np.sort(first)]] = np.arange(V)
        return ids
    ids_cm = class_major(prm)
    evaluate("cm-rmpoints", ids_cm, nbr_u, V)
    evaluate("cm-zpoints", class_major(zp), nbr_u, V)
    # ... and with the d+1 corners of a point taken in the order of their own codes
    code_rm = np.zeros(V, np.int64)
    stride = 1
    for j in range(d):
        code_rm += cs[:, j].astype(np.int64) * stride
        stride *= int(span[j])
    # which axis runs fastest
    for fast in range(d):
        rest = [j for j in range(d) if j != fast]
        evaluate("rm-fast%d" % fast, order_ids([cs[:, fast]] + [cs[:, j] for j in rest]), nbr_u, V)
    # rows along axis 0, the rows themselves along a Z-order curve of the other coordinates
    for bits in (15, 25):
        zc = zcode(cs[:, 1:], bits)
        evaluate("z%d(c1..5)+row0" % bits, order_ids([cs[:, 0], ids_zp * 0, zc]), nbr_u, V)
    # 2-D tiles: (c0, c1) row-major inside, Z-order of (c2..c5) outside
    zc = zcode(cs[:, 2:], 20)
    evaluate("h2", order_ids([cs[:, 0], cs[:, 1], zc]), nbr_u, V)                     # z20(c2..5) + row-major (c1, c0)
    zc = zcode(cs[:, 3:], 15)
    evaluate("h3", order_ids([cs[:, 0], cs[:, 1], cs[:, 2], zc]), nbr_u, V)           # z15(c3..5) + row-major (c2, c1, c0)
    zc = zcode(cs[:, 4:], 10)
    evaluate("h4", order_ids([cs[:, 0], cs[:, 1], cs[:, 2], cs[:, 3], zc]), nbr_u, V) # z10(c4..5) + row-major (c3 .. c0)
    # boustrophedon: every other row runs backwards (the end of a row is next to the start of the following one)
    par = (cs[:, 1:].sum(1) & 1)
    evaluate("snake", order_ids([np.where(par == 1, span[0] - 1 - cs[:, 0], cs[:, 0])] + [cs[:, j] for j in range(1, d)]), nbr_u, V)
    # slice / splat side: lines per gather of 64 consecutive POINTS reading their remainder-r vertex, by point order
    def slice_lines(name, ids, porder):
        off = ids[inv].reshape(N, D1)[porder]
        nw = (N + 63) // 64
        padn = nw * 64 - N
        tot = 0.0
        for r in range(D1):
            l = np.concatenate([(off[:, r] + 2) // 16, np.full(padn, -1)]).reshape(nw, 64)
            l = np.sort(l, axis=1)
            tot += (np.diff(l, axis=1) != 0).sum() + nw
        print("   slice: %-28s lines per gather instruction %.2f" % (name, tot / (D1 * nw)))
    # splat side: the Q gathers of 64 consecutive VERTICES' rows (first entry of every row: one gather instruction), by point order
    def splat_lines(name, ids, porder):
        ppos = np.empty(N, np.int64)
        ppos[porder] = np.arange(N)                 # original point -> internal position
        ent_v = ids[inv]                            # vertex id of every entry (point-major)
        order = np.lexsort((np.arange(N * D1), ent_v))
        first = np.ones(len(order), bool)
        first[1:] = ent_v[order][1:] != ent_v[order][:-1]
        fpt = ppos[order[first] // D1]              # internal position of the first point of every row, by vertex id
        nw = (V + 63) // 64
        l = np.concatenate([fpt // 16, np.full(nw * 64 - V, -1)]).reshape(nw, 64)
        l = np.sort(l, axis=1)
        print("   splat: %-28s lines per first-entry gather %.2f" % (name, ((np.diff(l, axis=1) != 0).sum() + nw) / nw))
    splat_lines("zpoints ids, zpoints order", ids_zp, zp)
    ids_rm0 = order_ids([cs[:, j] for j in range(d)])
    splat_lines("rowmajor ids, zpoints order", ids_rm0, zp)
    pm0 = ids_rm0[inv].reshape(N, D1).min(1)
    splat_lines("rowmajor ids, points by min id", ids_rm0, np.lexsort((ident, pm0)))
    pm1 = np.sort(ids_rm0[inv].reshape(N, D1), axis=1)[:, D1 // 2]
    splat_lines("rowmajor ids, points by median id", ids_rm0, np.lexsort((ident, pm1)))
    slice_lines("rowmajor ids, points by median id", ids_rm0, np.lexsort((ident, pm1)))
    slice_lines("zpoints ids, zpoints order", ids_zp, zp)
    for nm, ii in (("h2", order_ids([cs[:, 0], cs[:, 1], zcode(cs[:, 2:], 20)])), ("h3", order_ids([cs[:, 0], cs[:, 1], cs[:, 2], zcode(cs[:, 3:], 15)])),
                   ("h1", order_ids([cs[:, 0], zcode(cs[:, 1:], 25)])), ("zc30", order_ids([zcode(cs, 30)]))):
        slice_lines(nm + " ids, zpoints order", ii, zp)
        pm = ii[inv].reshape(N, D1).min(1)
        slice_lines(nm + " ids, points by min id", ii, np.lexsort((ident, pm)))
    slice_lines("cm-rmpoints ids, rmpoints order", ids_cm, prm)
    slice_lines("cm-zpoints ids, zpoints order", class_major(zp), zp)
    ids_rm = order_ids([cs[:, j] for j in range(d)])
    slice_lines("rowmajor ids, zpoints order", ids_rm, zp)
    pmin = ids_rm[inv].reshape(N, D1).min(1)
    slice_lines("rowmajor ids, points by min id", ids_rm, np.lexsort((ident, pmin)))
    c = zcode(vcells, 21)
    g = c ^ (c >> 1)
    # position along the Gray-code walk = inverse Gray transform of the Z code read as a Gray code
    b = c.copy()
    sh = 1
    while sh < 32:
        b ^= b >> sh
        sh <<= 1
    order = np.lexsort((ids_zp, b))
    ids = np.empty(V, np.int64)
    ids[order] = np.arange(V)
    evaluate("gray21", ids, nbr_u, V)


if __name__ == "__main__":
    main()
