#!/usr/bin/env python3
"""Writes the compiled early stages of a stump cascade: csrc/spec/spec_<tag>.inc, straight-line device code for stages
0 .. N-1 in which every LDS corner offset is an immediate of the ds_read, every threshold a literal and every leaf
difference a scalar constant.  The generic kernel forms each corner address with one VALU instruction and fetches the
node records by scalar loads; here a stump needs neither ("compiled early stages").

The input is a cascade in the package's own HCC1 container (tools/pack_cascade.py; ugproj-detector_amd/data/cascades/*.hcc -- data, the
same numbers as the XML).  The file records an FNV-1a hash of the model (api_model.cpp model_hash walks the very same fields
in the same order); the library takes the compiled stages only for a loaded cascade whose hash, window and tile
geometry match, anything else runs the generic kernel.  Only stages whose f64 sum is exact in any order are compiled
(same test as api_model.cpp stage_sum_is_exact) and only integer-weight cascades (weights_are_small_integers).  Cascades
of two-node trees (haarcascade_frontalface_alt2: a root, one of whose branches is a leaf) are compiled with both nodes of a
tree evaluated and the leaf chosen by two selects.

usage: gen_spec.py <cascade.hcc> <tag> <n stages> <out.inc> [band_rows_ystep1 band_rows_ystep2] [--p16 | --cv24]

--cv24: the stages with the node arithmetic of OpenCV 2.4 (cascade_body.inc HC_CV24, csrc/spec_unit_cv24.inc): hidden
area-normalised f32 rect weights as immediates, value = (double)((float)S0 * w0) + (double)((float)S1 * w1) [+ ..], node test
value < (double)thr * vnf with the f64 variance factor, stage bias 0.0001f.
--p16: the unit of the 16-bit-tile build of the kernel (cascade_body.inc HC_P16; scan step 1 only): corner offsets in
2-byte entries, rect sums modulo 2^16, and for the row-form stages 0-1 a PAIR form that evaluates two adjacent windows per
lane from 4-byte reads of entry pairs (even corner columns from copy A of the tile, odd ones from the shifted copy B).
A rect 0 of more than 257 pixels is cut in two (two-rect stumps) or replaced by its checkerboard quadrants (three-rect stumps);
refused when a compiled stage has a large rect of any other shape.
"""
import math
import struct
import sys

import numpy as np

import os

TILE_POS = 64
GROUP = 2        # stumps whose corner reads are issued together
CV24_GROUP = int(os.environ.get("HC_GEN_CV24_GROUP", "2"))   # ... in --cv24 units (f64 values and constants: twice the registers per stump)
DUP_READS = int(os.environ.get("HC_GEN_DUP_READS", "0"))   # experiments only: never set when the committed files are generated


def load_hcc(path):
    b = open(path, "rb").read()
    if b[:4] != b"HCC1":
        raise SystemExit("not an HCC1 file: " + path)
    win_w, win_h, nst, ntr, nnd, nlf, flags = struct.unpack_from("<7i", b, 4)
    p = 32
    st = np.frombuffer(b, dtype=np.dtype([("thr", "<f8"), ("nt", "<i4"), ("par", "<i4"), ("nxt", "<i4")]), count=nst, offset=p)
    p += nst * 20
    tn = np.frombuffer(b, "<i4", ntr, p)
    p += ntr * 4
    nd = np.frombuffer(b, dtype=np.dtype([("r", "i1", (3, 4)), ("w", "<f4", 3), ("thr", "<f4"), ("left", "<i4"),
                                          ("right", "<i4"), ("tilted", "<i4")]), count=nnd, offset=p)
    p += nnd * 40
    lf = np.frombuffer(b, "<f4", nlf, p)
    return dict(win_w=win_w, win_h=win_h, flags=flags, stages=st, tree_nnodes=tn, nodes=nd, leaves=lf)


def f32bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def f32val(bits):
    return struct.unpack("<f", struct.pack("<I", bits))[0]


def model_hash(m):
    """FNV-1a 64 over: win_w, win_h, nstages, then per stage ntrees and the f32 bits of the device threshold
    ((float)thr - 1e-5f in f32), the node count of every tree, per node 12 rect bytes (as u8), 3 weight words, thr word
    and the two child links, then every leaf word."""
    h = 0xcbf29ce484222325

    def feed(bs):
        nonlocal h
        for c in bs:
            h = ((h ^ c) * 0x100000001b3) & 0xffffffffffffffff

    feed(struct.pack("<3i", m["win_w"], m["win_h"], len(m["stages"])))
    for s in m["stages"]:
        thr = np.float32(np.float32(s["thr"]) - np.float32(1e-5))
        feed(struct.pack("<iI", int(s["nt"]), f32bits(thr)))
    for t in m["tree_nnodes"]:
        feed(struct.pack("<i", int(t)))
    for n in m["nodes"]:
        feed(bytes(int(v) & 0xff for v in n["r"].reshape(-1)))
        feed(struct.pack("<4I", f32bits(n["w"][0]), f32bits(n["w"][1]), f32bits(n["w"][2]), f32bits(n["thr"])))
        feed(struct.pack("<2i", int(n["left"]), int(n["right"])))
    for v in m["leaves"]:
        feed(struct.pack("<I", f32bits(v)))
    return h


def stage_exact(leaves):
    emin = None
    for v in leaves:
        v = float(v)
        if v == 0.0:
            continue
        if not math.isfinite(v):
            return False
        e = math.frexp(v)[1] - 24
        emin = e if emin is None else min(emin, e)
    if emin is None:
        return True
    units = 0
    for v in leaves:
        u = math.ldexp(abs(float(v)), -emin)
        if u > 2.0 ** 60:
            return False
        units += int(u)
    return 2 * units < 2 ** 53


def stump_corners(n):
    cs = set()
    for r in range(3):
        x, y, w, h = [int(v) for v in n["r"][r]]
        if r == 2 and int(n["w"][2]) == 0:
            continue
        cs |= {(x, y), (x + w, y), (x, y + h), (x + w, y + h)}
    return cs


def order_for_sharing(nodes, ks):
    """The stumps of a stage, reordered so that the stumps read together (GROUP at a time) share as many rect corners
    as a greedy choice finds: a shared corner is read once per group.  Legal because only stages whose f64 sum is exact in
    any order are compiled.  Deterministic (ties go to the lower index)."""
    cs = {k: stump_corners(nodes[k]) for k in ks}
    rem, out = list(ks), []
    while rem:
        grp = [rem.pop(0)]
        have = set(cs[grp[0]])
        while len(grp) < GROUP and rem:
            best = max(rem, key=lambda k: (len(cs[k] & have), -k))
            rem.remove(best)
            grp.append(best)
            have |= cs[best]
        out += grp
    return out


def geom(win_w, win_h, ystep, band_rows):
    cols = (TILE_POS - 1) * ystep + win_w + 1
    rows = (band_rows - 1) * ystep + win_h + 1
    if ystep == 1:
        return cols, 0, rows
    half = (cols + 1) // 2
    return 2 * half, half, rows


PAIR_STAGES = int(os.environ.get("HC_GEN_PAIR_STAGES", "2"))   # --p16: stages with a pair form (the kernel's row-form stages)


def emit_p16(L, m, nodes, first, nst, pitch, tree_terms, dwords, order):
    """stage code of the 16-bit-tile build, scan step 1: T16(a, byte offset) = one entry (zero-extended), T2(a, byte offset) =
    an aligned entry pair (hc_u16x2); `a` / `ab` = LDS byte address of the window origin in copy A / copy B."""
    def off(dx, dy):
        return (dy * pitch + dx) * 2

    def rects(n):
        """(coefficient, corners) of the feature value f = sum coefficient * rect sum, every rect of at most 257 pixels so that
        its sum modulo 2^16 is its sum.  Rect 0 (weight -1) of more than 257 pixels: in a two-rect stump it is cut in two
        halves (the same cut as api_model.cpp p16_model); in a three-rect stump it must be the 2 x 2 grid of a checkerboard
        feature -- rects 1 and 2 two diagonal quadrants with weight 2 -- and 2 (S1 + S2) - S0 = S1 + S2 - S3 - S4 with the
        other two quadrants S3, S4 (device: the checker form of cascade_body.inc)."""
        R = [(int(n["w"][r]), [int(v) for v in n["r"][r]]) for r in range(3) if not (r == 2 and int(n["w"][2]) == 0)]
        big = [i for i, (_, (x, y, w, h)) in enumerate(R) if w * h > 257]
        if not big:
            out = R
        elif big != [0] or R[0][0] != -1:
            raise SystemExit("--p16: a rect of more than 257 pixels that is not rect 0 with weight -1")
        elif len(R) == 2:
            x, y, w, h = R[0][1]
            if h >= w:
                h1 = h // 2
                out = [(-1, [x, y, w, h1]), (-1, [x, y + h1, w, h - h1]), R[1]]
            else:
                w1 = w // 2
                out = [(-1, [x, y, w1, h]), (-1, [x + w1, y, w - w1, h]), R[1]]
            if max(r[2] * r[3] for _, r in out) > 257:
                # halves still too large (24 x 24 windows: rect 0 of up to 576 pixels).  Rect 1 spans rect 0's full height or
                # width: w1 S1 - S0 = (w1 - 1) S1 - (the one or two strips of rect 0 beside rect 1)
                wt, (x1, y1, w1, h1) = R[1]
                if (y1, h1) == (y, h) and x <= x1 and x1 + w1 <= x + w:
                    strips = [[x, y, x1 - x, h], [x1 + w1, y, x + w - x1 - w1, h]]
                elif (x1, w1) == (x, w) and y <= y1 and y1 + h1 <= y + h:
                    strips = [[x, y, w, y1 - y], [x, y1 + h1, w, y + h - y1 - h1]]
                else:
                    raise SystemExit("--p16: a two-rect stump whose rect 0 neither halves nor complements into rects of <= 257 pixels")
                out = [(-1, st) for st in strips if st[2] * st[3] > 0] + [(wt - 1, [x1, y1, w1, h1])]
        else:
            (x, y, w, h), (x1, y1, w1, h1), (x2, y2, w2, h2) = R[0][1], R[1][1], R[2][1]
            if R[1][0] != 2 or R[2][0] != 2:
                raise SystemExit("--p16: three-rect stump with a large rect 0 and weights other than (-1, 2, 2)")
            if (x1, y1) == (x, y) and (x2 + w2, y2 + h2) == (x + w, y + h) and (x1 + w1, y1 + h1) == (x2, y2):      # diagonal
                xm, ym = x2, y2
                out = [(1, R[1][1]), (1, R[2][1]), (-1, [xm, y, x + w - xm, ym - y]), (-1, [x, ym, xm - x, y + h - ym])]
            elif (x1 + w1, y1) == (x + w, y) and (x2, y2 + h2) == (x, y + h) and (x1, y1 + h1) == (x2 + w2, y2):    # anti-diagonal
                xm, ym = x1, y1 + h1
                out = [(1, R[1][1]), (1, R[2][1]), (-1, [x, y, xm - x, ym - y]), (-1, [xm, ym, x + w - xm, y + h - ym])]
            else:
                raise SystemExit("--p16: three-rect stump with a large rect 0 that is not a checkerboard")
            if sum(r[2] * r[3] for _, r in out) != w * h:
                raise SystemExit("--p16: checkerboard quadrants do not tile rect 0")
        for _, (x, y, w, h) in out:
            if w * h > 257 or w <= 0 or h <= 0:
                raise SystemExit("--p16: a rect of more than 257 pixels remains")
        return [(c, [(x, y), (x + w, y), (x, y + h), (x + w, y + h)]) for c, (x, y, w, h) in out]

    def combine(terms):
        """terms: (coefficient, C expression of a rect sum) -> the feature value as a C expression"""
        pos = [e if c == 1 else "__mul24(%d, %s)" % (c, e) for c, e in terms if c > 0]
        neg = [e if c == -1 else "__mul24(%d, %s)" % (-c, e) for c, e in terms if c < 0]
        return (" + ".join(pos) if pos else "0") + "".join(" - " + e for e in neg)

    def feature1(n, corner):
        terms = []
        for coef, cs in rects(n):
            c = [corner.setdefault(off(dx, dy), "c%d" % len(corner)) for dx, dy in cs]
            terms.append((coef, "((%s - %s - %s + %s) & 0xffff)" % tuple(c)))
        return combine(terms)

    def feature2(n, corner, sums):
        """two windows: rect sums as hc_u16x2 (modulo 2^16 by construction), then one feature per window"""
        names = []
        for coef, cs in rects(n):
            c = []
            for dx, dy in cs:
                key = ("A", off(dx, dy)) if dx % 2 == 0 else ("B", off(dx - 1, dy))
                c.append(corner.setdefault(key, "p%d" % len(corner)))
            nm = "s%d" % len(sums)
            sums.append("const hc_u16x2 %s = %s - %s - %s + %s;" % ((nm,) + tuple(c)))
            names.append((coef, nm))
        return combine([(c, "(int)%s.x" % nm) for c, nm in names]), combine([(c, "(int)%s.y" % nm) for c, nm in names])

    for s in range(nst):
        thr = np.float32(np.float32(m["stages"]["thr"][s]) - np.float32(1e-5))
        bsum = sum(tree_terms(t)[0] for t in range(first[s], first[s + 1]))
        ks = order(nodes, list(range(first[s], first[s + 1])))
        L.append("  static __device__ __forceinline__ bool stage%d_y1(int a, float vnf) {" % s)
        L.append("    double acc = %s;" % float(bsum).hex())
        for g0 in range(0, len(ks), GROUP):
            grp = ks[g0:g0 + GROUP]
            corner, feats, consts = {}, [], []
            for k in grp:
                feats.append(feature1(nodes[k], corner))
                consts.append((f32bits(nodes[k]["thr"]),) + dwords(tree_terms(k)[1][0]))
            L.append("    {")
            for gi, (_, lo, hi) in enumerate(consts):
                L.append("      int l%d, h%d; K(l%d, h%d, 0x%08x, 0x%08x);" % (gi, gi, gi, gi, lo, hi))
            L.append("      const int " + ", ".join("%s = T16(a, %d)" % (nm, o) for o, nm in corner.items()) + ";")
            L.append("      __builtin_amdgcn_sched_barrier(0);")
            for gi, f in enumerate(feats):
                L.append("      const int f%d = %s;" % (gi, f))
            for gi, (thrb, _, _) in enumerate(consts):
                L.append("      A(acc, vnf, f%d, 0x%08xu, l%d, h%d);" % (gi, thrb, gi, gi))
            L.append("    }")
        L.append("    return !(acc < (double)__uint_as_float(0x%08xu));" % f32bits(thr))
        L.append("  }")
        if s >= PAIR_STAGES:
            continue
        L.append("  static __device__ __forceinline__ void pair%d_y1(int a, int ab, float v0, float v1, bool& ok0, bool& ok1) {" % s)
        L.append("    double acc0 = %s, acc1 = acc0;" % float(bsum).hex())
        for g0 in range(0, len(ks), GROUP):
            grp = ks[g0:g0 + GROUP]
            corner, sums, feats, consts = {}, [], [], []
            for k in grp:
                feats.append(feature2(nodes[k], corner, sums))
                consts.append((f32bits(nodes[k]["thr"]),) + dwords(tree_terms(k)[1][0]))
            L.append("    {")
            for gi, (_, lo, hi) in enumerate(consts):
                L.append("      int l%d, h%d; K(l%d, h%d, 0x%08x, 0x%08x);" % (gi, gi, gi, gi, lo, hi))
            L.append("      const hc_u16x2 " + ", ".join("%s = T2(%s, %d)" % (nm, "a" if cp == "A" else "ab", o) for (cp, o), nm in corner.items()) + ";")
            L.append("      __builtin_amdgcn_sched_barrier(0);")
            for ln in sums:
                L.append("      " + ln)
            for gi, (fx, fy) in enumerate(feats):
                L.append("      const int f%da = %s, f%db = %s;" % (gi, fx, gi, fy))
            for gi, (thrb, _, _) in enumerate(consts):
                L.append("      A(acc0, v0, f%da, 0x%08xu, l%d, h%d);" % (gi, thrb, gi, gi))
                L.append("      A(acc1, v1, f%db, 0x%08xu, l%d, h%d);" % (gi, thrb, gi, gi))
            L.append("    }")
        L.append("    ok0 = !(acc0 < (double)__uint_as_float(0x%08xu));" % f32bits(thr))
        L.append("    ok1 = !(acc1 < (double)__uint_as_float(0x%08xu));" % f32bits(thr))
        L.append("  }")
    L.append("  template <int YS> static __device__ __forceinline__ bool eval(int s, int a, float vnf) {")
    L.append("    switch (s) {")
    for s in range(nst):
        L.append("      case %d: return stage%d_y1(a, vnf);" % (s, s))
    L.append("      default: return false;")
    L.append("    }")
    L.append("  }")
    L.append("  template <int YS> static __device__ __forceinline__ void eval_pair(int s, int a, int ab, float v0, float v1, bool& ok0, bool& ok1) {")
    L.append("    switch (s) {")
    for s in range(min(PAIR_STAGES, nst)):
        L.append("      case %d: pair%d_y1(a, ab, v0, v1, ok0, ok1); break;" % (s, s))
    L.append("      default: ok0 = ok1 = false;")
    L.append("    }")
    L.append("  }")


def main():
    argv = [a for a in sys.argv if a not in ("--p16", "--cv24")]
    p16 = "--p16" in sys.argv
    cv24 = "--cv24" in sys.argv          # OpenCV 2.4 node arithmetic (cascade_body.inc HC_CV24)
    if p16 and cv24:
        raise SystemExit("--p16 and --cv24 exclude each other")
    path, tag, nst, out = argv[1], argv[2], int(argv[3]), argv[4]
    band_rows = [int(argv[5]), int(argv[6])] if len(argv) > 6 else [16, 16]
    m = load_hcc(path)
    tn = m["tree_nnodes"]
    if int(tn.max()) == 1:
        is_tree = False
    elif int(tn.min()) == 2 and int(tn.max()) == 2:
        is_tree = True                  # every tree: a root, one of whose branches is a leaf and the other the second node
    else:
        raise SystemExit("neither a stump cascade nor one of two-node trees")
    nodes, leaves = m["nodes"], m["leaves"]
    for n in nodes:
        if n["w"][0] != -1.0 or any(float(w) != round(float(w)) or abs(float(w)) > 127 for w in n["w"]):
            raise SystemExit("weights are not small integers")
    first = np.concatenate([[0], np.cumsum(m["stages"]["nt"])]).astype(int)        # first tree of a stage
    nfirst = np.concatenate([[0], np.cumsum(tn)]).astype(int)                        # first node of a tree
    lfirst = np.concatenate([[0], np.cumsum(tn + 1)]).astype(int)                    # first leaf of a tree

    def tree_terms(t):
        """(constant leaf, [(leaf difference, ...)]) of tree t such that leaf = constant + sum(selected differences):
        stump: right + [v < thr] * (left - right); two nodes: other + go * ((lb - other) + [v1 < thr1] * (la - lb))"""
        n0 = nodes[nfirst[t]]
        lf = leaves[lfirst[t]:lfirst[t + 1]]
        if not is_tree:
            return float(lf[1]), [float(lf[0]) - float(lf[1])], None
        n1 = nodes[nfirst[t] + 1]
        l, r = int(n0["left"]), int(n0["right"])
        if (l > 0) == (r > 0) or max(l, r) != 1 or int(n1["left"]) > 0 or int(n1["right"]) > 0:
            raise SystemExit("tree %d is not root + one inner child" % t)
        left_is_node = l > 0
        other = float(lf[-r]) if left_is_node else float(lf[-l])
        la, lb = float(lf[-int(n1["left"])]), float(lf[-int(n1["right"])])
        return other, [lb - other, la - lb], left_is_node

    def stage_exact_terms(s):
        """every partial sum of constant + selected differences is exact in f64 (all terms multiples of 2^emin, the sum of
        their magnitudes below 2^53) -- and so is upstream's ordered sum of the leaves themselves"""
        lv = leaves[lfirst[first[s]]:lfirst[first[s + 1]]]
        if not stage_exact(lv):
            return False
        emin = min(math.frexp(float(v))[1] - 24 for v in lv if float(v) != 0.0) if any(float(v) != 0.0 for v in lv) else 0
        units = 0
        for t in range(first[s], first[s + 1]):
            c, ds, _ = tree_terms(t)
            for v in [c] + ds:
                u = math.ldexp(abs(v), -emin)
                if u != int(u):
                    return False
                units += int(u)
        return units < 2 ** 53

    nst = min(nst, len(m["stages"]))
    for s in range(nst):
        if not stage_exact_terms(s):
            nst = s
            break

    def dwords(d):
        bits = struct.unpack("<Q", struct.pack("<d", d))[0]
        return bits & 0xffffffff, bits >> 32
    L = []
    L.append("// GENERATED by tools/gen_spec.py from %s -- do not edit.  Stages 0..%d of the cascade as straight-line code." % (path.split("/")[-1], nst - 1))
    L.append("// Included inside namespace hc::%s by csrc/%s, after cascade_body.inc's helpers." %
             (("casc_cv24", "spec_unit_cv24.inc") if cv24 else ("casc_p16", "spec_unit_p16.inc") if p16 else ("casc_int", "spec_unit.inc")))
    L.append("struct Spec_%s {" % tag)
    L.append("  static constexpr unsigned long long hash = 0x%016xull;" % model_hash(m))
    L.append("  static constexpr int win_w = %d, win_h = %d, nstages = %d;" % (m["win_w"], m["win_h"], nst))
    L.append("  static constexpr bool is_tree = %s;   // two nodes per tree (the kernel's TREE instantiation)" % ("true" if is_tree else "false"))
    L.append("  static constexpr int band_rows1 = %d, band_rows2 = %d;" % (band_rows[0], band_rows[1]))
    if cv24:
        L.append("  static constexpr bool cv24 = true;   // 2.4 node arithmetic: hidden f32 rect weights, f64 value, value < thr * vnf (f64)")
    if p16:
        if is_tree:
            raise SystemExit("--p16: stump cascades only")
        pitch16 = (geom(m["win_w"], m["win_h"], 1, band_rows[0])[0] + 1) & ~1
        L.append("  static constexpr int pitch1 = %d, pitch2 = 0;   // LDS tile row pitch in 2-byte entries (scan step 1 only)" % pitch16)
        L.append("  static constexpr bool p16 = true;")
        L.append("  static constexpr int pair_stages = %d;" % min(PAIR_STAGES, nst))
        emit_p16(L, m, nodes, first, nst, pitch16, tree_terms, dwords, order_for_sharing)
        L.append("};")
        open(out, "w").write("\n".join(L) + "\n")
        print("%s: %d stages (16-bit tile, %d with a pair form), hash %016x" % (tag, nst, min(PAIR_STAGES, nst), model_hash(m)))
        return
    L.append("  static constexpr int pitch1 = %d, pitch2 = %d;   // LDS tile row pitch (words) the offsets below assume" %
             (geom(m["win_w"], m["win_h"], 1, band_rows[0])[0], geom(m["win_w"], m["win_h"], 2, band_rows[1])[0]))

    VNF_T, F_T = ("double", "double") if cv24 else ("float", "int")
    for ys in (1, 2):
        pitch, half, _ = geom(m["win_w"], m["win_h"], ys, band_rows[ys - 1])

        def off(dx, dy):
            col = dx if ys == 1 else ((dx & 1) * half + (dx >> 1))
            return (dy * pitch + col) * 4

        def hidden_weights(n):
            """B.6, the operations of legacy_cv24.cpp cv24_weights in their types: w_k = (float)(weight_k * inv_area) for the rects
            1 and 2 (f64 product), rect 0 balances them, w_0 = (float)(-sum_k (w_k * width_k * height_k) / area_0) with the
            products formed in f32 and summed in f64"""
            inv_area = 1.0 / float((m["win_w"] - 2) * (m["win_h"] - 2))
            w = [np.float32(0), np.float32(0), np.float32(0)]
            sum0 = 0.0
            for k in (1, 2):
                if float(n["w"][k]) == 0.0:
                    continue
                w[k] = np.float32(float(n["w"][k]) * inv_area)
                sum0 += float(np.float32(np.float32(w[k] * np.float32(int(n["r"][k][2]))) * np.float32(int(n["r"][k][3]))))
            w[0] = np.float32(-sum0 / float(int(n["r"][0][2]) * int(n["r"][0][3])))
            return w

        def feature(n, corner):
            terms = []
            hw = hidden_weights(n) if cv24 else None
            for r in range(3):
                x, y, w, h = [int(v) for v in n["r"][r]]
                wt = int(n["w"][r])
                if r == 2 and wt == 0:
                    continue
                c = [corner.setdefault(o, "c%d" % len(corner)) for o in (off(x, y), off(x + w, y), off(x, y + h), off(x + w, y + h))]
                if cv24:        # (double)((float)S_r * w_r), added in rect order
                    terms.append("(double)((float)(%s - %s - %s + %s) * __uint_as_float(0x%08xu))" % (c[0], c[1], c[2], c[3], f32bits(hw[r])))
                elif r == 0:
                    terms.append("(%s + %s - %s - %s)" % (c[1], c[2], c[0], c[3]))
                else:
                    terms.append("__mul24(%d, %s - %s - %s + %s)" % (wt, c[0], c[1], c[2], c[3]))
            if cv24:
                return " + ".join(terms)
            return " + ".join(terms[1:] + terms[:1])

        for s in range(nst):
            thr = np.float32(np.float32(m["stages"]["thr"][s]) - np.float32(1e-4 if cv24 else 1e-5))   # B.8: stage bias 0.0001f
            bsum = sum(tree_terms(t)[0] for t in range(first[s], first[s + 1]))   # exact in f64 by stage_exact_terms
            L.append("  static __device__ __forceinline__ bool stage%d_y%d(int a, %s vnf) {" % (s, ys, VNF_T))
            L.append("    double acc = %s;" % float(bsum).hex())
            if is_tree:
                for t in range(first[s], first[s + 1]):                # one tree (two nodes) per read batch
                    n0, n1 = nodes[nfirst[t]], nodes[nfirst[t] + 1]
                    _, (d1, d2), left_is_node = tree_terms(t)
                    corner = {}
                    f0, f1 = feature(n0, corner), feature(n1, corner)
                    L.append("    {")
                    L.append("      int l0, h0; K(l0, h0, 0x%08x, 0x%08x);" % dwords(d1))
                    L.append("      int l1, h1; K(l1, h1, 0x%08x, 0x%08x);" % dwords(d2))
                    if cv24:
                        L.append("      int tl0, th0; K(tl0, th0, 0x%08x, 0x%08x);" % dwords(float(np.float32(n0["thr"]))))
                        L.append("      int tl1, th1; K(tl1, th1, 0x%08x, 0x%08x);" % dwords(float(np.float32(n1["thr"]))))
                    L.append("      const int " + ", ".join("%s = T(a, %d)" % (nm, o) for o, nm in corner.items()) + ";")
                    L.append("      __builtin_amdgcn_sched_barrier(0);")
                    L.append("      const %s f0 = %s;" % (F_T, f0))
                    L.append("      const %s f1 = %s;" % (F_T, f1))
                    if cv24:
                        L.append("      A2(acc, vnf, f0, tl0, th0, f1, tl1, th1, %s, l0, h0, l1, h1);" % ("true" if left_is_node else "false"))
                    else:
                        L.append("      A2(acc, vnf, f0, 0x%08xu, f1, 0x%08xu, %s, l0, h0, l1, h1);" %
                                 (f32bits(n0["thr"]), f32bits(n1["thr"]), "true" if left_is_node else "false"))
                    L.append("    }")
            else:
                ks = order_for_sharing(nodes, list(range(first[s], first[s + 1])))
                grp_n = CV24_GROUP if cv24 else GROUP
                for g0 in range(0, len(ks), grp_n):
                    grp = ks[g0:g0 + grp_n]
                    corner = {}                      # byte offset -> local name (corners shared inside the group are read once)
                    feats, consts = [], []
                    for gi, k in enumerate(grp):
                        n = nodes[k]
                        feats.append(feature(n, corner))
                        consts.append((f32bits(n["thr"]),) + dwords(tree_terms(k)[1][0]))
                    L.append("    {")
                    for gi, (thrb, lo, hi) in enumerate(consts):
                        L.append("      int l%d, h%d; K(l%d, h%d, 0x%08x, 0x%08x);" % (gi, gi, gi, gi, lo, hi))
                        if cv24:    # (double)thr as a 64-bit constant set where it is used (an f64 operand cannot be a literal)
                            L.append("      int tl%d, th%d; K(tl%d, th%d, 0x%08x, 0x%08x);" % ((gi, gi, gi, gi) + dwords(float(f32val(thrb)))))
                    L.append("      const int " + ", ".join("%s = T(a, %d)" % (nm, o) for o, nm in corner.items()) + ";")
                    if s < DUP_READS:       # sensitivity probe (HC_GEN_DUP_READS): every corner of the first stages read twice
                        for o in corner:
                            L.append("      asm volatile(\"\" :: \"v\"(T(a, %d)));" % o)
                    L.append("      __builtin_amdgcn_sched_barrier(0);")
                    for gi, f in enumerate(feats):
                        L.append("      const %s f%d = %s;" % (F_T, gi, f))
                    for gi, (thrb, _, _) in enumerate(consts):
                        if cv24:
                            L.append("      A(acc, vnf, f%d, tl%d, th%d, l%d, h%d);" % (gi, gi, gi, gi, gi))
                        else:
                            L.append("      A(acc, vnf, f%d, 0x%08xu, l%d, h%d);" % (gi, thrb, gi, gi))
                    L.append("    }")
            L.append("    return !(acc < (double)__uint_as_float(0x%08xu));" % f32bits(thr))
            L.append("  }")
        L.append("  static __device__ __forceinline__ bool eval_y%d(int s, int a, %s vnf) {" % (ys, VNF_T))
        L.append("    switch (s) {")
        for s in range(nst):
            L.append("      case %d: return stage%d_y%d(a, vnf);" % (s, s, ys))
        L.append("      default: return false;")
        L.append("    }")
        L.append("  }")
    L.append("  template <int YS> static __device__ __forceinline__ bool eval(int s, int a, %s vnf) {" % VNF_T)
    L.append("    return YS == 1 ? eval_y1(s, a, vnf) : eval_y2(s, a, vnf);")
    L.append("  }")
    L.append("};")
    open(out, "w").write("\n".join(L) + "\n")
    print("%s: %d stages, %d %s compiled, hash %016x" % (tag, nst, first[nst], "two-node trees" if is_tree else "stumps", model_hash(m)))


if __name__ == "__main__":
    main()
