"""Second, independent restatement of the path in numpy (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED.  Purpose: the C++ oracle (oracle/haar_oracle.cpp) must not
be self-confirming.  This module restates the same OpenCV>=4 semantics for the call at
reference src/tracker/detector.cc:25-31 with a *different structure*:
  - dense evaluation of every grid position, all windows of a level at once (array ops),
  - the x-skip through its closed form instead of the sequential loop,
  - resize coefficients through exact rational arithmetic cross-checked against the f64 rule.
It is slow (small images only).  Only tests/ may import it.
"""
import math
import struct

import numpy as np

F32 = np.float32
F64 = np.float64


def cv_round(v):
    """cvRound: round half to even (np.rint) on the value's own precision."""
    return int(np.rint(v))


# ----------------------------------------------------------------------------- model
class Cascade:
    pass


def load_hcc(path):
    b = open(path, "rb").read()
    assert b[:4] == b"HCC1"
    win_w, win_h, nst, ntr, nnd, nlf, flags = struct.unpack_from("<7i", b, 4)
    p = 32
    c = Cascade()
    c.win_w, c.win_h, c.tilted = win_w, win_h, flags & 1
    st = np.frombuffer(b, dtype=np.dtype([("thr", "<f8"), ("nt", "<i4"), ("par", "<i4"), ("nxt", "<i4")]), count=nst, offset=p)
    p += nst * 20
    c.stage_thr_raw = st["thr"].copy()
    # A.2: (float)t - 1e-5f evaluated in f32
    c.stage_thr = (st["thr"].astype(F32) - F32(1e-5)).astype(F32)
    c.stage_ntrees = st["nt"].copy()
    c.stage_parent = st["par"].copy()
    c.stage_next = st["nxt"].copy()
    c.tree_nnodes = np.frombuffer(b, "<i4", ntr, p).copy()
    p += ntr * 4
    nd = np.frombuffer(b, dtype=np.dtype([("r", "i1", (3, 4)), ("w", "<f4", 3), ("thr", "<f4"), ("left", "<i4"),
                                          ("right", "<i4"), ("tilted", "<i4")]), count=nnd, offset=p)
    p += nnd * 40
    c.rects = nd["r"].astype(np.int32)
    c.weights = nd["w"].copy()
    c.node_thr = nd["thr"].copy()
    c.node_left = nd["left"].copy()
    c.node_right = nd["right"].copy()
    c.leaves = np.frombuffer(b, "<f4", nlf, p).copy()
    c.max_nodes = int(c.tree_nnodes.max())
    return c


# ----------------------------------------------------------------------------- plan (A.3, A.6b)
def plan(c, rows, cols, sf=1.1, min_size=(0, 0), max_size=(0, 0)):
    min_w, min_h = min_size
    max_w, max_h = max_size
    if max_w == 0 or max_h == 0:
        max_w, max_h = cols, rows
    if rows < c.win_h or cols < c.win_w:
        return [], 0
    all_scales = []
    factor = 1.0
    while True:
        ww, wh = cv_round(c.win_w * factor), cv_round(c.win_h * factor)
        if ww > cols or wh > rows:
            break
        all_scales.append(F32(factor))
        factor *= sf
    scales = []
    for s in all_scales:
        ww, wh = cv_round(F32(c.win_w) * s), cv_round(F32(c.win_h) * s)
        if ww > max_w or wh > max_h:
            break
        if ww < min_w or wh < min_h:
            continue
        scales.append(s)
    if not scales and all_scales:
        d = [(min_w - cv_round(F32(c.win_w) * s)) ** 2 + (min_h - cv_round(F32(c.win_h) * s)) ** 2 for s in all_scales]
        scales.append(all_scales[int(np.argmin(d))])
    levels = []
    for sc in scales:
        w = cv_round(F32(cols) / sc)
        h = cv_round(F32(rows) / sc)
        ystep = 1 if sc >= 2 else 2
        levels.append(dict(scale=sc, w=w, h=h, ystep=ystep, szw_w=max(w + 1 - c.win_w, 0), szw_h=max(h + 1 - c.win_h, 0),
                           win_w=cv_round(F32(c.win_w) * sc), win_h=cv_round(F32(c.win_h) * sc)))
    if not levels:
        return [], 0
    nstripes = int(math.ceil(levels[0]["szw_w"] / 32.0))
    for l in levels:
        l["stripe"] = max((l["szw_h"] // l["ystep"] + nstripes - 1) // max(nstripes, 1), 1) * l["ystep"]
        l["y_end"] = min(nstripes * l["stripe"], l["szw_h"])
    return levels, nstripes


# ----------------------------------------------------------------------------- resize (A.4)
def lin_coeffs(src, dst):
    """(ofs, c1) per destination index; replicate zones folded in as c1 = 0 with a clamped ofs.

    fval = scale*(d+0.5)-0.5 with scale = 1/(dst/src) in f64.  The fraction is ALSO computed in
    exact rational arithmetic ((2d+1)*src - dst)/(2*dst) and the test-suite asserts that both
    roundings agree for the ladder sizes (they can differ only if the f64 product lands on the
    other side of a .5/256 tie)."""
    inv_scale = F64(dst) / F64(src)
    scale = F64(1.0) / inv_scale
    d = np.arange(dst, dtype=F64)
    fval = scale * (d + 0.5) - 0.5
    ival = np.floor(fval).astype(np.int64)
    ofs = np.zeros(dst, np.int64)
    c1 = np.zeros(dst, np.int64)
    if src > 1:
        main = (ival >= 0) & (ival < src - 1)
        ofs[main] = ival[main]
        c1[main] = np.rint((fval[main] - ival[main]) * 256.0).astype(np.int64)
        right = ival >= src - 1
        ofs[right] = src - 1
    return ofs, c1


def lin_coeffs_rational(src, dst):
    """Exact-arithmetic version of the ideal coefficients (no f64 rounding of scale)."""
    ofs, c1 = [], []
    for d in range(dst):
        num = (2 * d + 1) * src - dst          # fval = num / (2*dst)
        den = 2 * dst
        iv = num // den
        if iv >= 0 and src > 1:
            if iv < src - 1:
                fr_num = (num - iv * den) * 256   # frac*256 = fr_num/den
                q, r = divmod(fr_num, den)
                if 2 * r > den or (2 * r == den and (q & 1)):
                    q += 1
                ofs.append(iv)
                c1.append(q)
            else:
                ofs.append(src - 1)
                c1.append(0)
        else:
            ofs.append(0)
            c1.append(0)
    return np.array(ofs, np.int64), np.array(c1, np.int64)


def resize_linear_exact(img, drows, dcols):
    rows, cols = img.shape
    if (rows, cols) == (drows, dcols):
        return img.copy()
    ox, cx1 = lin_coeffs(cols, dcols)
    oy, cy1 = lin_coeffs(rows, drows)
    cx0, cy0 = 256 - cx1, 256 - cy1
    s = img.astype(np.int64)
    ox1 = np.minimum(ox + 1, cols - 1)
    oy1 = np.minimum(oy + 1, rows - 1)
    h = s[:, ox] * cx0[None, :] + s[:, ox1] * cx1[None, :]         # 8.8 fixed point, <= 65280
    v = h[oy, :] * cy0[:, None] + h[oy1, :] * cy1[:, None]          # 16.16
    return np.minimum((v + 32768) >> 16, 255).astype(np.uint8)


# ----------------------------------------------------------------------------- integral (A.5)
def integral(img):
    rows, cols = img.shape
    v = img.astype(np.uint64)
    s = np.zeros((rows + 1, cols + 1), np.uint64)
    q = np.zeros((rows + 1, cols + 1), np.uint64)
    s[1:, 1:] = np.cumsum(np.cumsum(v, axis=0), axis=1)
    q[1:, 1:] = np.cumsum(np.cumsum(v * v, axis=0), axis=1)
    return (s & 0xFFFFFFFF).astype(np.uint32), (q & 0xFFFFFFFF).astype(np.uint32)   # modulo 2^32


# ----------------------------------------------------------------------------- dense evaluation
def _rect_sum(I, xs, ys, r):
    x, y, w, h = (int(t) for t in r)
    a = I[np.ix_(ys + y, xs + x)]
    b = I[np.ix_(ys + y, xs + x + w)]
    cc = I[np.ix_(ys + y + h, xs + x)]
    d = I[np.ix_(ys + y + h, xs + x + w)]
    return (a - b - cc + d).astype(np.uint32).view(np.int32)        # wraps like int32


def _feature(c, S, xs, ys, ni, vnf):
    w = c.weights[ni]
    ret = w[0] * _rect_sum(S, xs, ys, c.rects[ni, 0]).astype(F32)
    ret = (ret + w[1] * _rect_sum(S, xs, ys, c.rects[ni, 1]).astype(F32)).astype(F32)
    if w[2] != 0:
        ret = (ret + w[2] * _rect_sum(S, xs, ys, c.rects[ni, 2]).astype(F32)).astype(F32)
    return (ret * vnf).astype(F32)


def dense_stage_pass(c, S, Q, level):
    """For every grid position of a level: -1 not evaluable, else number of stages passed
    *if evaluated* (no skip logic here)."""
    ys_ = np.arange(0, level["szw_h"], level["ystep"])
    xs_ = np.arange(0, level["szw_w"], level["ystep"])
    if len(ys_) == 0 or len(xs_) == 0:
        return np.zeros((len(ys_), len(xs_)), np.int32)
    nr = (1, 1, c.win_w - 2, c.win_h - 2)
    valsum = _rect_sum(S, xs_, ys_, nr).astype(F64)
    valsq = _rect_sum(Q, xs_, ys_, nr).view(np.uint32).astype(F64)
    area = F64((c.win_w - 2) * (c.win_h - 2))
    nf = area * valsq - valsum * valsum
    pos = nf > 0
    vnf = np.ones(nf.shape, F32)
    vnf[pos] = (1.0 / np.sqrt(nf[pos])).astype(F32)
    evaluable = pos & (area * vnf.astype(F64) < 1e-1)
    nst = len(c.stage_thr)
    passed = np.full(nf.shape, nst, np.int32)
    alive = evaluable.copy()
    node0 = 0
    leaf0 = 0
    tree = 0
    for si in range(nst):
        tmp = np.zeros(nf.shape, F64)
        for _ in range(int(c.stage_ntrees[si])):
            nn = int(c.tree_nnodes[tree])
            if c.max_nodes == 1:
                v = _feature(c, S, xs_, ys_, node0, vnf).astype(F64)
                tmp += np.where(v < F64(c.node_thr[node0]), F64(c.leaves[leaf0]), F64(c.leaves[leaf0 + 1]))
            else:
                idx = np.zeros(nf.shape, np.int32)
                leafv = np.zeros(nf.shape, F64)
                done = np.zeros(nf.shape, bool)
                for _depth in range(nn + 1):
                    for k in range(nn):
                        m = (~done) & (idx == k)
                        if not m.any():
                            continue
                        v = _feature(c, S, xs_, ys_, node0 + k, vnf).astype(F64)
                        nxt = np.where(v < F64(c.node_thr[node0 + k]), c.node_left[node0 + k], c.node_right[node0 + k])
                        idx = np.where(m, nxt, idx)
                        fin = m & (nxt <= 0)
                        leafv = np.where(fin, c.leaves[leaf0 - np.minimum(nxt, 0)].astype(F64), leafv)
                        done |= fin
                    if done.all():
                        break
                tmp += leafv
            node0 += nn
            leaf0 += nn + 1
            tree += 1
        fail = alive & (tmp < F64(c.stage_thr[si]))
        passed[fail] = si
        alive &= ~fail
    passed[~evaluable] = -1
    return passed


def apply_skip(passed, level):
    """Closed form of A.7 on a dense 'passed' grid: returns taps with -2 for unvisited."""
    ny_all, nx_all = passed.shape
    taps = np.full(passed.shape, -2, np.int8)
    ny = (level["y_end"] + level["ystep"] - 1) // level["ystep"]
    for gy in range(min(ny, ny_all)):
        f = passed[gy] == 0                      # evaluable and fails stage 0 if evaluated
        # anchors: k == 0 or not f[k-1]; visited iff distance to nearest anchor at/left is even
        anchor = np.ones(nx_all, bool)
        anchor[1:] = ~f[:-1]
        idx = np.arange(nx_all)
        last_anchor = np.maximum.accumulate(np.where(anchor, idx, 0))
        vis = ((idx - last_anchor) % 2) == 0
        taps[gy, vis] = passed[gy, vis]
    return taps


# ----------------------------------------------------------------------------- grouping (A.10)
def _similar(a, b, eps):
    delta = eps * (min(a[2], b[2]) + min(a[3], b[3])) * 0.5
    return (abs(a[0] - b[0]) <= delta and abs(a[1] - b[1]) <= delta and
            abs(a[0] + a[2] - b[0] - b[2]) <= delta and abs(a[1] + a[3] - b[1] - b[3]) <= delta)


def group_rectangles(rects, thr, eps=0.2):
    rects = [tuple(int(v) for v in r) for r in rects]
    if thr <= 0 or not rects:
        return rects
    n = len(rects)
    # connected components by BFS (independent of the union-find in the C++ oracle)
    label = [-1] * n
    ncls = 0
    for i in range(n):
        if label[i] >= 0:
            continue
        label[i] = ncls
        stack = [i]
        while stack:
            a = stack.pop()
            for j in range(n):
                if label[j] < 0 and _similar(rects[a], rects[j], eps):
                    label[j] = ncls
                    stack.append(j)
        ncls += 1
    sums = np.zeros((ncls, 4), np.int64)
    cnt = np.zeros(ncls, np.int64)
    for r, k in zip(rects, label):
        sums[k] += r
        cnt[k] += 1
    mean = []
    for k in range(ncls):
        s = F32(1.0) / F32(cnt[k])
        mean.append(tuple(cv_round(F32(v) * s) for v in sums[k]))
    out = []
    for i in range(ncls):
        r1, n1 = mean[i], cnt[i]
        if n1 <= thr:
            continue
        keep = True
        for j in range(ncls):
            n2 = cnt[j]
            if j == i or n2 <= thr:
                continue
            r2 = mean[j]
            dx, dy = cv_round(r2[2] * eps), cv_round(r2[3] * eps)
            if (r1[0] >= r2[0] - dx and r1[1] >= r2[1] - dy and r1[0] + r1[2] <= r2[0] + r2[2] + dx and
                    r1[1] + r1[3] <= r2[1] + r2[3] + dy and (n2 > max(3, n1) or n1 < 3)):
                keep = False
                break
        if keep:
            out.append(r1)
    return out


# ----------------------------------------------------------------------------- whole call
def detect(c, gray, sf=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), want_taps=False):
    rows, cols = gray.shape
    levels, _ = plan(c, rows, cols, sf, min_size, max_size)
    raw, taps_all = [], []
    nst = len(c.stage_thr)
    for l in levels:
        img = resize_linear_exact(gray, l["h"], l["w"])
        S, Q = integral(img)
        passed = dense_stage_pass(c, S, Q, l)
        taps = apply_skip(passed, l)
        taps_all.append(taps)
        gy, gx = np.nonzero(taps == nst)
        for y, x in zip(gy, gx):          # row-major == scale -> y -> x
            xx, yy = int(x) * l["ystep"], int(y) * l["ystep"]
            raw.append((cv_round(F32(xx) * l["scale"]), cv_round(F32(yy) * l["scale"]), l["win_w"], l["win_h"]))
    grouped = group_rectangles(raw, min_neighbors, 0.2)
    if want_taps:
        return grouped, raw, taps_all
    return grouped, raw


# ----------------------------------------------------------------------------- adjacent ops
def bgr2gray(img):
    """cvtColor(BGR2GRAY), 8-bit: the 15-bit coefficient set (gray_shift = 15)."""
    a = np.asarray(img).astype(np.int64)
    return ((a[..., 0] * 3735 + a[..., 1] * 19235 + a[..., 2] * 9798 + 16384) >> 15).astype(np.uint8)


def bgr2ycrcb_y(img):
    """Y channel of cvtColor(BGR2YCrCb), 8-bit: the 14-bit coefficient set (yuv_shift = 14)."""
    a = np.asarray(img).astype(np.int64)
    return (a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + 8192) >> 14


def _area_weights(ssize, dsize):
    """Dense [dsize, ssize] f32 weight matrix + per-row tap order of INTER_AREA's general path.  Written from the
    cell-coverage definition (how much of source cell s lies inside destination cell d, over the cell width) with the
    1e-3 edge rule of computeResizeAreaTab, not from the tap loop of the C++ oracle."""
    scale = F64(ssize) / F64(dsize)
    W = np.zeros((dsize, ssize), F32)
    order = []
    for d in range(dsize):
        lo, hi = F64(d) * scale, F64(d) * scale + scale
        cell = min(scale, F64(ssize) - lo)
        first_full, end_full = math.ceil(lo), min(math.floor(hi), ssize - 1)
        first_full = min(first_full, end_full)
        taps = []
        if first_full - lo > 1e-3:
            W[d, first_full - 1] = F32((first_full - lo) / cell)
            taps.append(first_full - 1)
        for s in range(first_full, end_full):
            W[d, s] = F32(1.0 / cell)
            taps.append(s)
        if hi - end_full > 1e-3:
            W[d, end_full] = F32(min(min(hi - end_full, 1.0), cell) / cell)
            taps.append(end_full)
        order.append(taps)
    return W, order


def resize_area(img, drows, dcols):
    """cv::resize(INTER_AREA) for shrinking 8-bit images (HxW or HxWxC), integer and non-integer ratios."""
    a = np.asarray(img)
    rows, cols = a.shape[:2]
    if a.ndim == 2:
        a = a[:, :, None]
    ch = a.shape[2]
    if rows % drows == 0 and cols % dcols == 0:
        fy, fx = rows // drows, cols // dcols
        blk = a.astype(np.int64).reshape(drows, fy, dcols, fx, ch).sum(axis=(1, 3))
        out = (blk + 2) >> 2 if (fx, fy) == (2, 2) else np.rint(blk.astype(F32) * (F32(1) / F32(fx * fy))).astype(np.int64)
    else:
        Wx, ox = _area_weights(cols, dcols)
        Wy, oy = _area_weights(rows, drows)
        out = np.zeros((drows, dcols, ch), np.int64)
        src = a.astype(F32)
        # horizontal pass: buf[sy, dx, c] accumulated tap by tap in f32 (left to right)
        buf = np.zeros((rows, dcols, ch), F32)
        for dx in range(dcols):
            acc = np.zeros((rows, ch), F32)
            for sx in ox[dx]:
                acc = (acc + src[:, sx, :] * Wx[dx, sx]).astype(F32)
            buf[:, dx, :] = acc
        for dy in range(drows):
            acc = None
            for sy in oy[dy]:
                t = (Wy[dy, sy] * buf[sy]).astype(F32)
                acc = t if acc is None else (acc + t).astype(F32)
            out[dy] = np.rint(acc).astype(np.int64)
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if np.asarray(img).ndim == 2 else out


# ----------------------------------------------------------------------------- T2: OpenCV 2.4.x legacy semantics
# Second restatement of oracle/haar_oracle.cpp's orc_detect_cv24 (rules B.1-B.9 there), array-at-a-time: every window
# of a level is evaluated together, breadth-first over the stage tree (all windows that currently sit at stage s
# take stage s in one go), instead of the C++ oracle's window-at-a-time walk.
def plan_cv24(c, rows, cols, sf=1.1, min_size=(0, 0), max_size=(0, 0)):
    max_w, max_h = max_size if (max_size[0] and max_size[1]) else (cols, rows)
    levels, factor = [], 1.0
    while True:
        ww, wh = cv_round(c.win_w * factor), cv_round(c.win_h * factor)
        w, h = cv_round(cols / factor), cv_round(rows / factor)
        if w - c.win_w + 1 <= 0 or h - c.win_h + 1 <= 0 or ww > max_w or wh > max_h:
            break
        if ww >= min_size[0] and wh >= min_size[1]:
            ystep = 1 if factor > 2 else 2
            levels.append(dict(factor=factor, w=w, h=h, ystep=ystep, nx=-(-(w - c.win_w) // ystep), ny=-(-(h - c.win_h) // ystep),
                               win_w=ww, win_h=wh))
        factor *= sf
    return levels


def _lin24(ssize, dsize):
    d = np.arange(dsize, dtype=F64)
    scale = F64(1.0) / (F64(dsize) / F64(ssize))
    f = ((d + 0.5) * scale - 0.5).astype(F32)
    sidx = np.floor(f).astype(np.int64)
    f = (f - sidx.astype(F32)).astype(F32)
    lo, hi = sidx < 0, sidx >= ssize - 1
    f[lo | hi] = 0
    sidx = np.clip(sidx, 0, ssize - 1)
    a1 = np.rint(f * F32(2048)).astype(np.int64)
    a0 = np.rint((F32(1) - f) * F32(2048)).astype(np.int64)
    return sidx, np.minimum(sidx + 1, ssize - 1), a0, a1


def resize_linear_cv24(img, drows, dcols):
    a = np.asarray(img).astype(np.int64)
    rows, cols = a.shape
    if (rows, cols) == (drows, dcols):
        return np.asarray(img).copy()
    if rows == 2 * drows and cols == 2 * dcols:
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, sx1, ax0, ax1 = _lin24(cols, dcols)
    sy, sy1, ay0, ay1 = _lin24(rows, drows)
    hor = a[:, sx] * ax0[None, :] + a[:, sx1] * ax1[None, :]                 # every source row, destination columns
    v = ((ay0[:, None] * (hor[sy] >> 4)) >> 16) + ((ay1[:, None] * (hor[sy1] >> 4)) >> 16)
    return ((v + 2) >> 2).astype(np.uint8)


def detect_cv24(c, gray, sf=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), want_taps=False):
    rows, cols = gray.shape
    inv_area = F64(1.0) / F64((c.win_w - 2) * (c.win_h - 2))
    # hidden weights (B.6) and stage thresholds / children (B.7, B.8)
    rects = c.rects.astype(np.int64)
    area = rects[:, :, 2] * rects[:, :, 3]
    w = (c.weights.astype(F64) * inv_area).astype(F32)
    w[c.weights == 0] = 0
    sum0 = (w[:, 1].astype(F64) * area[:, 1]) + 0.0
    three = c.weights[:, 2] != 0
    sum0 = np.where(three, sum0 + w[:, 2].astype(F64) * area[:, 2], sum0)
    w[:, 0] = (-sum0 / area[:, 0]).astype(F32)
    stage_thr = (c.stage_thr_raw.astype(F32) - F32(0.0001)).astype(F32)
    nst = len(stage_thr)
    child = np.full(nst, -1)
    for i in range(nst):
        pnt = int(c.stage_parent[i])
        if 0 <= pnt < nst and child[pnt] < 0:
            child[pnt] = i
    tree_first = np.concatenate([[0], np.cumsum(c.stage_ntrees)])
    node_first = np.concatenate([[0], np.cumsum(c.tree_nnodes)])
    leaf_first = np.concatenate([[0], np.cumsum(c.tree_nnodes + 1)])
    cands, taps = [], []
    for L in plan_cv24(c, rows, cols, sf, min_size, max_size):
        img = resize_linear_cv24(gray, L["h"], L["w"])
        S = np.zeros((L["h"] + 1, L["w"] + 1), np.int64)
        S[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
        Q = np.zeros_like(S)
        Q[1:, 1:] = (img.astype(np.int64) ** 2).cumsum(0).cumsum(1)
        gy, gx = np.mgrid[0:L["ny"], 0:L["nx"]]
        y, x = (gy * L["ystep"]).ravel(), (gx * L["ystep"]).ravel()

        def box(I, r, idx):
            yy, xx = y[idx] + r[1], x[idx] + r[0]
            return I[yy, xx] - I[yy, xx + r[2]] - I[yy + r[3], xx] + I[yy + r[3], xx + r[2]]
        eq = (1, 1, c.win_w - 2, c.win_h - 2)
        allw = np.arange(len(x))
        mean = box(S, eq, allw).astype(F64) * inv_area
        var = box(Q, eq, allw).astype(F64) * inv_area - mean * mean
        vnf = np.where(var >= 0, np.sqrt(np.maximum(var, 0)), 1.0)
        at = np.zeros(len(x), np.int64)            # stage each window sits at; -1 rejected, -2 accepted
        passed = np.zeros(len(x), np.int64)
        while (at >= 0).any():
            for s in np.unique(at[at >= 0]):
                idx = np.nonzero(at == s)[0]
                ssum = np.zeros(len(idx), F64)
                for t in range(tree_first[s], tree_first[s + 1]):
                    node = np.zeros(len(idx), np.int64)
                    leaf = np.zeros(len(idx), F64)
                    live = np.ones(len(idx), bool)
                    while live.any():
                        for k in np.unique(node[live]):
                            sel = live & (node == k)
                            ni = node_first[t] + k
                            thr = F64(c.node_thr[ni]) * vnf[idx[sel]]
                            fs = (box(S, rects[ni, 0], idx[sel]).astype(F32) * w[ni, 0]).astype(F64)
                            fs = fs + (box(S, rects[ni, 1], idx[sel]).astype(F32) * w[ni, 1]).astype(F64)
                            if c.weights[ni, 2] != 0:
                                fs = fs + (box(S, rects[ni, 2], idx[sel]).astype(F32) * w[ni, 2]).astype(F64)
                            go = np.where(fs < thr, c.node_left[ni], c.node_right[ni])
                            sub = np.nonzero(sel)[0]
                            done = go <= 0
                            leaf[sub[done]] = c.leaves[leaf_first[t] - go[done]].astype(F64)
                            live[sub[done]] = False
                            node[sub[~done]] = go[~done]
                    ssum = ssum + leaf
                ok = ~(ssum < F64(stage_thr[s]))
                passed[idx[ok]] += 1
                nxt = np.empty(len(idx), np.int64)
                nxt[ok] = child[s] if child[s] >= 0 else -2
                # failure: next sibling of the nearest ancestor-or-self that has one
                f = s
                while f >= 0 and c.stage_next[f] < 0:
                    f = int(c.stage_parent[f])
                nxt[~ok] = int(c.stage_next[f]) if f >= 0 else -1
                at[idx] = nxt
        acc = at == -2
        taps.append((np.minimum(passed, 63) | np.where(acc, 64, 0)).astype(np.int8).reshape(L["ny"], L["nx"]))
        for i in np.nonzero(acc)[0]:
            cands.append((cv_round(x[i] * L["factor"]), cv_round(y[i] * L["factor"]), L["win_w"], L["win_h"]))
    raw = np.array(cands, np.int32).reshape(-1, 4)
    grouped = group_rectangles(raw, max(min_neighbors, 1)) if min_neighbors != 0 else raw
    return (grouped, raw, taps) if want_taps else (grouped, raw)
