"""ctypes wrapper of oracle/libhaar_oracle.so -- the CPU oracle (TEST INFRASTRUCTURE ONLY).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.  The
product (ugproj-detector_amd/) never touches this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libhaar_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "ugproj-detector_amd", "data")     # the packed cascades ship with the product package
CASCADE_NAMES = ("alt", "alt2", "alt_tree", "default")


def cascade_path(name, xml=False):
    if xml:
        return "/root/reference/data/haarcascade_frontalface_%s.xml" % name
    return os.path.join(DATA, "cascades", "haarcascade_frontalface_%s.hcc" % name)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


class Level(C.Structure):
    _fields_ = [("scale", C.c_float)] + [(n, C.c_int32) for n in
                                         "w h ystep szw_w szw_h stripe y_end nx ny win_w win_h".split()]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src_m = max(os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("haar_oracle.cpp", "haar_oracle.h"))
        if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < src_m:
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_cascade_load_xml.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.orc_cascade_load_hcc.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.orc_cascade_free.argtypes = [C.c_void_p]
        L.orc_cascade_free.restype = None
        L.orc_cascade_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_cascade_dump.argtypes = [C.c_void_p] * 9
        L.orc_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_int] * 4 + \
                              [C.POINTER(Level), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_resize_linear_exact.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.orc_integral.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_group_rectangles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                           C.POINTER(C.c_int)]
        L.orc_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double] + [C.c_int] * 5 + \
                                [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                 C.c_void_p, C.c_size_t]
        L.orc_detect_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double] + [C.c_int] * 6 + \
                                     [C.c_void_p]
        L.orc_eval_patch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_resize_linear_cv24.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.orc_plan_cv24.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_int,
                                                                                              C.POINTER(C.c_int)]
        L.orc_detect_cv24.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double] + [C.c_int] * 5 + \
                                     [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                      C.c_void_p, C.c_size_t]
        L.orc_bgr2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
        L.orc_resize_area.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
        L.orc_detect_faces.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.c_double, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class OracleCascade:
    def __init__(self, path):
        self._h = C.c_void_p()
        fn = lib().orc_cascade_load_hcc if path.endswith(".hcc") else lib().orc_cascade_load_xml
        rc = fn(os.fsencode(path), C.byref(self._h))
        if rc:
            raise RuntimeError("oracle cannot load %s (%d)" % (path, rc))
        info = (C.c_int32 * 8)()
        lib().orc_cascade_info(self._h, info)
        (self.win_w, self.win_h, self.nstages, self.ntrees, self.nnodes, self.nleaves, self.max_nodes,
         self.tilted) = list(info)

    def __del__(self):
        try:
            if self._h:
                lib().orc_cascade_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def dump(self):
        st = np.zeros(self.nstages, np.float32)
        sn = np.zeros(self.nstages, np.int32)
        tn = np.zeros(self.ntrees, np.int32)
        nr = np.zeros(self.nnodes * 15, np.int32)
        nt = np.zeros(self.nnodes, np.float32)
        nl = np.zeros(self.nnodes, np.int32)
        nrr = np.zeros(self.nnodes, np.int32)
        lf = np.zeros(self.nleaves, np.float32)
        lib().orc_cascade_dump(self._h, *[a.ctypes.data for a in (st, sn, tn, nr, nt, nl, nrr, lf)])
        return st, sn, tn, nr, nt, nl, nrr, lf

    def plan(self, rows, cols, sf=1.1, min_size=(0, 0), max_size=(0, 0)):
        buf = (Level * 256)()
        n, ns = C.c_int(), C.c_int()
        rc = lib().orc_plan(self._h, rows, cols, sf, min_size[0], min_size[1], max_size[0], max_size[1], buf, 256,
                            C.byref(n), C.byref(ns))
        if rc:
            raise RuntimeError("orc_plan %d" % rc)
        return [buf[i] for i in range(n.value)], ns.value

    def tap_sizes(self, levels):
        return [((l.szw_h + l.ystep - 1) // l.ystep, (l.szw_w + l.ystep - 1) // l.ystep) for l in levels]

    def detect(self, gray, sf=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), want_taps=False,
               cap=1 << 14, raw_cap=1 << 20):
        """-> grouped (K,4), raw (N,4) [, taps: list of int8 [gy][gx] per level]"""
        gray = np.ascontiguousarray(gray, np.uint8)
        rows, cols = gray.shape
        out = np.zeros((cap, 4), np.int32)
        raw = np.zeros((raw_cap, 4), np.int32)
        n, nr = C.c_int(), C.c_int()
        taps, tp, tcap, shapes = None, None, 0, []
        if want_taps:
            levels, _ = self.plan(rows, cols, sf, min_size, max_size)
            shapes = self.tap_sizes(levels)
            taps = np.zeros(sum(a * b for a, b in shapes), np.int8)
            tp, tcap = taps.ctypes.data, taps.size
        rc = lib().orc_detect(self._h, gray.ctypes.data, rows, cols, gray.strides[0], sf, min_neighbors, min_size[0],
                              min_size[1], max_size[0], max_size[1], out.ctypes.data, cap, C.byref(n), raw.ctypes.data,
                              raw_cap, C.byref(nr), tp, tcap)
        if rc:
            raise RuntimeError("orc_detect %d" % rc)
        res = [out[:n.value].copy(), raw[:nr.value].copy()]
        if want_taps:
            per, o = [], 0
            for a, b in shapes:
                per.append(taps[o:o + a * b].reshape(a, b))
                o += a * b
            res.append(per)
        return res

    def plan_cv24(self, rows, cols, sf=1.1, min_size=(0, 0), max_size=(0, 0)):
        """levels of the OpenCV-2.4 (T2) call: list of dicts factor, w, h, ystep, nx, ny, win_w, win_h"""
        fac = np.zeros(256, np.float64)
        geo = np.zeros((256, 7), np.int32)
        n = C.c_int()
        rc = lib().orc_plan_cv24(self._h, rows, cols, sf, min_size[0], min_size[1], max_size[0], max_size[1], fac.ctypes.data,
                                 geo.ctypes.data, 256, C.byref(n))
        if rc:
            raise RuntimeError("orc_plan_cv24 %d" % rc)
        keys = "w h ystep nx ny win_w win_h".split()
        return [dict(factor=float(fac[i]), **{k: int(v) for k, v in zip(keys, geo[i])}) for i in range(n.value)]

    def detect_cv24(self, gray, sf=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), want_taps=False, cap=1 << 16,
                    raw_cap=1 << 20):
        """T2 / OpenCV 2.4.x semantics -> grouped, raw [, taps per level: stages passed | 64 if accepted]"""
        gray = np.ascontiguousarray(gray, np.uint8)
        rows, cols = gray.shape
        out = np.zeros((cap, 4), np.int32)
        raw = np.zeros((raw_cap, 4), np.int32)
        n, nr = C.c_int(), C.c_int()
        levels = self.plan_cv24(rows, cols, sf, min_size, max_size)
        taps = np.zeros(max(sum(l["nx"] * l["ny"] for l in levels), 1), np.int8)
        rc = lib().orc_detect_cv24(self._h, gray.ctypes.data, rows, cols, gray.strides[0], sf, min_neighbors, min_size[0],
                                   min_size[1], max_size[0], max_size[1], out.ctypes.data, cap, C.byref(n), raw.ctypes.data,
                                   raw_cap, C.byref(nr), taps.ctypes.data if want_taps else None, taps.size)
        if rc:
            raise RuntimeError("orc_detect_cv24 %d" % rc)
        res = [out[:n.value].copy(), raw[:nr.value].copy()]
        if want_taps:
            per, o = [], 0
            for l in levels:
                per.append(taps[o:o + l["nx"] * l["ny"]].reshape(l["ny"], l["nx"]))
                o += l["nx"] * l["ny"]
            res.append(per)
        return res

    def detect_faces(self, bgr, lower=(0, 133, 77), upper=(255, 173, 127), scale=1.1, want_taps=False, cap=4096):
        """FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43) -> rects [, mask, gray]"""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        rows, cols, _ = bgr.shape
        lo = (C.c_int * 3)(*lower)
        hi = (C.c_int * 3)(*upper)
        out = np.zeros((cap, 4), np.int32)
        n = C.c_int()
        mask = np.zeros((rows, cols), np.uint8)
        gray = np.zeros((rows, cols), np.uint8)
        rc = lib().orc_detect_faces(self._h, bgr.ctypes.data, rows, cols, bgr.strides[0], lo, hi, scale, out.ctypes.data,
                                    cap, C.byref(n), mask.ctypes.data, gray.ctypes.data)
        if rc:
            raise RuntimeError("orc_detect_faces %d" % rc)
        return (out[:n.value].copy(), mask, gray) if want_taps else out[:n.value].copy()

    def detect_many(self, frames, sf=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), nthreads=1):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, rows, cols = frames.shape
        counts = np.zeros(n, np.int32)
        rc = lib().orc_detect_many(self._h, frames.ctypes.data, n, rows, cols, sf, min_neighbors, min_size[0],
                                   min_size[1], max_size[0], max_size[1], nthreads, counts.ctypes.data)
        if rc:
            raise RuntimeError("orc_detect_many %d" % rc)
        return counts


def resize_linear_exact(img, drows, dcols):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros((drows, dcols), np.uint8)
    rc = lib().orc_resize_linear_exact(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], dst.ctypes.data,
                                       drows, dcols)
    assert rc == 0
    return dst


def resize_linear_cv24(img, drows, dcols):
    img = np.ascontiguousarray(img, np.uint8)
    dst = np.zeros((drows, dcols), np.uint8)
    rc = lib().orc_resize_linear_cv24(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], dst.ctypes.data, drows, dcols)
    assert rc == 0
    return dst


def bgr2gray(img):
    """cvtColor(BGR2GRAY) of an HxWx3 / HxWx4 uint8 image"""
    img = np.ascontiguousarray(img, np.uint8)
    g = np.zeros(img.shape[:2], np.uint8)
    rc = lib().orc_bgr2gray(img.ctypes.data, img.shape[0], img.shape[1], img.shape[2], img.strides[0], g.ctypes.data)
    if rc:
        raise RuntimeError("orc_bgr2gray %d" % rc)
    return g


def resize_area(img, drows, dcols):
    """cv::resize(..., INTER_AREA), shrinking in both axes; img HxW or HxWx3 uint8"""
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    dst = np.zeros((drows, dcols) if ch == 1 else (drows, dcols, ch), np.uint8)
    rc = lib().orc_resize_area(img.ctypes.data, img.shape[0], img.shape[1], ch, img.strides[0], dst.ctypes.data, drows, dcols)
    if rc:
        raise RuntimeError("orc_resize_area %d" % rc)
    return dst


def integral(img):
    img = np.ascontiguousarray(img, np.uint8)
    s = np.zeros((img.shape[0] + 1, img.shape[1] + 1), np.int32)
    q = np.zeros_like(s)
    lib().orc_integral(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], s.ctypes.data, q.ctypes.data)
    return s, q


def group_rectangles(rects, thr, eps=0.2):
    rects = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(-1, 4))
    out = np.zeros((max(len(rects), 1), 4), np.int32)
    n = C.c_int()
    rc = lib().orc_group_rectangles(rects.ctypes.data, len(rects), thr, eps, out.ctypes.data, len(out), C.byref(n))
    assert rc == 0
    return out[:n.value].copy()


def sorted_boxes(a):
    a = np.asarray(a, np.int32).reshape(-1, 4)
    if len(a) == 0:
        return a
    return a[np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))]
