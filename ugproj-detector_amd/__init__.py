"""ctypes binding of libhip_cascade.so (C ABI: include/hip_cascade.h).

Host-side mirror, for tests and bench.py, of the two calls the reference makes on its
cv::CascadeClassifier (reference src/file_io.cc:11 `load`, src/tracker/detector.cc:25-31
`detectMultiScale`).  The reference is C++, so the *product* host layer is the C++ shim
in hip_cascade/cascade_classifier.hpp; this module only exposes the same C ABI to Python.

There is no fallback of any kind: if the shared library is missing, import fails loudly;
if no HIP device is present, every compute call raises HipCascadeError(HC_ERR_NO_DEVICE).

The directory name contains a hyphen, so load this package with `load_package()` from
tests/conftest.py / bench.py (importlib by path) under the module name `ugproj_detector_amd`.
"""
import ctypes as C
import os

import numpy as np

try:
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  Whoever loads first wins the SONAME, and a
    # process must never hold two HIP runtimes: import torch (if present) BEFORE our library so
    # both bind the same one and device pointers can be shared (bench.py, device-input tests).
    import torch  # noqa: F401
except ImportError:      # pure C++/ctypes use: the system ROCm runtime is used
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# HC_LIB_PATH: A/B experiments with alternative builds of the same library (tools/exp_*.sh); never a different backend
LIB_PATH = os.environ.get("HC_LIB_PATH") or os.path.join(_HERE, "libhip_cascade.so")

DATA_DIR = os.path.join(_HERE, "data")
CASCADE_NAMES = ("alt", "alt2", "alt_tree", "default")


def cascade_path(name):
    """packed parameters of reference data/haarcascade_frontalface_<name>.xml that ship with the package"""
    if name not in CASCADE_NAMES:
        raise ValueError("unknown stock cascade %r (one of %s); pass a path to Cascade() for any other file" % (name, ", ".join(CASCADE_NAMES)))
    return os.path.join(DATA_DIR, "cascades", "haarcascade_frontalface_%s.hcc" % name)


HC_NKERNELS = 5
TAP_PYRAMID, TAP_SUM, TAP_SQSUM, TAP_VNF, TAP_STAGES, TAP_FACE_MASK, TAP_FACE_GRAY, TAP_FACE_BGR = range(8)
SKIN_LOWER, SKIN_UPPER = (0, 133, 77), (255, 173, 127)      # reference src/structure.cc:75-82 defaults

STATUS = {0: "HC_OK", -1: "HC_ERR_IO", -2: "HC_ERR_FORMAT", -3: "HC_ERR_ARG", -4: "HC_ERR_UNSUPPORTED",
          -5: "HC_ERR_NO_DEVICE", -6: "HC_ERR_HIP", -7: "HC_ERR_CAPACITY", -8: "HC_ERR_CAND_OVERFLOW",
          -9: "HC_ERR_STATE"}


class HipCascadeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class Level(C.Structure):
    _fields_ = [("scale", C.c_float), ("w", C.c_int32), ("h", C.c_int32), ("ystep", C.c_int32),
                ("szw_w", C.c_int32), ("szw_h", C.c_int32), ("y_end", C.c_int32), ("nx", C.c_int32),
                ("ny", C.c_int32), ("win_w", C.c_int32), ("win_h", C.c_int32), ("int_pitch", C.c_int32),
                ("pyr_ofs", C.c_int64), ("int_ofs", C.c_int64), ("grid_ofs", C.c_int64)]


class LaunchInfo(C.Structure):
    """hc_launch_info: what the last enqueue of a context launched"""
    _fields_ = [("prep_route", C.c_int32), ("variance_route", C.c_int32),
                ("pi_frame_chunk", C.c_int32), ("pi_remapped", C.c_int32), ("pi_tail_frames", C.c_int32),
                ("casc_frame_chunk", C.c_int32), ("casc_remapped", C.c_int32), ("casc_tail_frames", C.c_int32),
                ("variant", C.c_int32 * 2), ("compiled_stages", C.c_int32 * 2), ("segments", C.c_int32),
                ("side_stream", C.c_int32), ("sq_stored", C.c_int32), ("nframes", C.c_int32),
                ("band_step", C.c_int32 * 2), ("lds_bytes", C.c_int32 * 2), ("device_grouping", C.c_int32)]


VARIANT_NAMES = ("none", "f32", "int", "spec", "p16", "cv24", "cv24_tree", "cv24_window", "cv24_chains")       # hc_variant
PREP_NAMES = ("none", "fused", "parallel", "shared_follower", "cv24_fused", "cv24_parallel")   # hc_prep_route
VARIANCE_NAMES = ("none", "kernel", "ring", "readback", "leader", "ring_strips", "cv24_kernel")   # hc_variance_route


if not os.path.exists(LIB_PATH):
    raise ImportError("libhip_cascade.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C ugproj-detector_amd` -- there is no fallback implementation" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

# every symbol declared in include/hip_cascade.h (tests/test_abi.py checks the header against this list)
_SIGS = {
    "hc_cascade_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "hc_cascade_free": (None, [C.c_void_p]),
    "hc_context_stream": (C.c_void_p, [C.c_void_p]),
    "hc_cascade_info": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 4),
    "hc_cascade_counts": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_int)] * 3),
    "hc_cascade_compiled_stages": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "hc_spec_plugin_load": (C.c_int, [C.c_char_p]),
    "hc_context_compiled_stages": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hc_context_last_launch": (C.c_int, [C.c_void_p, C.POINTER(LaunchInfo)]),
    "hc_debug_plan_strips": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_int] * 6 +
                             [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "hc_cascade_dump": (C.c_int, [C.c_void_p] + [C.c_void_p] * 8),
    "hc_detect_multiscale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.POINTER(C.c_int)]),
    "hc_detect_multiscale_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                             C.POINTER(C.c_int)]),
    "hc_detect_multiscale_reread": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "hc_debug_alloc_count": (C.c_longlong, []),
    "hc_context_set_rerun_on_overflow": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_context_set_device_grouping": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_context_group_host_frames": (C.c_int, [C.c_void_p]),
    "hc_debug_enqueue_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hc_context_set_semantics": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_convert_bgr2gray": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]),
    "hc_boxes_save": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hc_boxes_load": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                                C.POINTER(C.c_int)]),
    "hc_detect_multiscale_cv24": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                            C.POINTER(C.c_int)]),
    "hc_detect_multiscale_cv24_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                 C.POINTER(C.c_int), C.c_void_p, C.c_size_t]),
    "hc_plan_query_cv24": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_int] * 4 +
                           [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "hc_resize_linear_cv24": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_int]),
    "hc_context_create": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "hc_context_destroy": (None, [C.c_void_p]),
    "hc_context_configure": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_int] * 7),
    "hc_context_configure_shared": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_double] + [C.c_int] * 7),
    "hc_detect_batch_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "hc_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "hc_host_free": (None, [C.c_void_p]),
    "hc_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "hc_host_unregister": (C.c_int, [C.c_void_p]),
    "hc_detect_batch_enqueue_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "hc_detect_batch_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hc_batch_raw_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hc_detect_faces": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                  C.c_double, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "hc_context_enable_faces": (C.c_int, [C.c_void_p]),
    "hc_context_enable_scan_resize": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "hc_faces_batch_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]),
    "hc_faces_batch_enqueue_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                              C.c_void_p]),
    "hc_faces_batch_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "hc_context_levels": (C.c_int, [C.c_void_p, C.POINTER(Level), C.c_int, C.POINTER(C.c_int)]),
    "hc_context_algorithmic_bytes": (C.c_int64, [C.c_void_p]),
    "hc_plan_query": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_int] * 4 +
                      [C.POINTER(Level), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "hc_group_rectangles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_int)]),
    "hc_context_enable_stage_taps": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_context_read_tap": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "hc_context_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "hc_context_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "hc_kernel_name": (C.c_char_p, [C.c_int]),
    "hc_last_error": (C.c_char_p, []),
    "hc_version": (C.c_char_p, []),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)      # AttributeError here == the ABI lost a symbol
    _fn.restype = _res
    _fn.argtypes = _args
EXPORTED = sorted(_SIGS)


def _check(rc):
    if rc != 0:
        raise HipCascadeError(rc, lib.hc_last_error().decode())


class Cascade:
    """cv::CascadeClassifier stand-in: load() + detectMultiScale()."""

    def __init__(self, path):
        self._h = C.c_void_p()
        _check(lib.hc_cascade_load(os.fsencode(path), C.byref(self._h)))
        v = [C.c_int() for _ in range(4)]
        _check(lib.hc_cascade_info(self._h, *[C.byref(x) for x in v]))
        self.win_w, self.win_h, self.nstages, self.is_tree = [x.value for x in v]
        v = [C.c_int() for _ in range(3)]
        _check(lib.hc_cascade_counts(self._h, *[C.byref(x) for x in v]))
        self.ntrees, self.nnodes, self.nleaves = [x.value for x in v]

    def close(self):
        if self._h:
            lib.hc_cascade_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compiled_stages(self):
        """(model hash, number of leading stages the library holds generated straight-line code for); pure host"""
        h, n = C.c_uint64(), C.c_int()
        _check(lib.hc_cascade_compiled_stages(self._h, C.byref(h), C.byref(n)))
        return h.value, n.value

    def dump(self):
        st = np.zeros(self.nstages, np.float32)
        sn = np.zeros(self.nstages, np.int32)
        tn = np.zeros(self.ntrees, np.int32)
        nr = np.zeros(self.nnodes * 15, np.int32)
        nt = np.zeros(self.nnodes, np.float32)
        nl = np.zeros(self.nnodes, np.int32)
        nrr = np.zeros(self.nnodes, np.int32)
        lf = np.zeros(self.nleaves, np.float32)
        _check(lib.hc_cascade_dump(self._h, *[a.ctypes.data for a in (st, sn, tn, nr, nt, nl, nrr, lf)]))
        return st, sn, tn, nr, nt, nl, nrr, lf

    def plan(self, rows, cols, scale_factor=1.1, min_size=(0, 0), max_size=(0, 0)):
        """(levels, algorithmic_bytes) of the call on a rows x cols image; pure host."""
        buf = (Level * 256)()
        n = C.c_int()
        ab = C.c_int64()
        _check(lib.hc_plan_query(self._h, rows, cols, scale_factor, min_size[0], min_size[1], max_size[0],
                                 max_size[1], buf, 256, C.byref(n), C.byref(ab)))
        return [buf[i] for i in range(n.value)], ab.value

    def plan_strips(self, rows, cols, scale_factor=1.1, min_size=(0, 0), max_size=(0, 0), cv24=False, comp_max=0):
        """(strips, lds_bytes): the column strips of the fused prep kernel for this call (hc_debug_plan_strips); strips is an
        (n, 12) int32 array: level, c0, c1, ce, g0, g1, sx0, sxn, left, has_right, carry_ofs, 0.  Pure host."""
        buf = np.zeros((4096, 12), np.int32)
        n, lds = C.c_int(), C.c_int64()
        _check(lib.hc_debug_plan_strips(self._h, rows, cols, scale_factor, min_size[0], min_size[1], max_size[0], max_size[1],
                                        1 if cv24 else 0, comp_max, buf.ctypes.data, len(buf), C.byref(n), C.byref(lds)))
        return buf[:n.value].copy(), lds.value

    def plan_cv24(self, rows, cols, scale_factor=1.1, min_size=(0, 0), max_size=(0, 0)):
        """levels of the legacy (OpenCV 2.4) call: list of dicts factor, w, h, ystep, nx, ny, win_w, win_h; pure host"""
        fac = np.zeros(256, np.float64)
        geo = np.zeros((256, 7), np.int32)
        n = C.c_int()
        _check(lib.hc_plan_query_cv24(self._h, rows, cols, scale_factor, min_size[0], min_size[1], max_size[0], max_size[1],
                                      fac.ctypes.data, geo.ctypes.data, 256, C.byref(n)))
        keys = "w h ystep nx ny win_w win_h".split()
        return [dict(factor=float(fac[i]), **{k: int(v) for k, v in zip(keys, geo[i])}) for i in range(n.value)]

    def detect_multiscale_cv24(self, gray, scale_factor=1.1, min_neighbors=3, flags=2, min_size=(0, 0), max_size=(0, 0),
                               capacity=1 << 16, want_taps=False):
        """legacy mode: OpenCV 2.4.x semantics (hc_detect_multiscale_cv24) -> (K,4) [, taps per level]"""
        gray = np.asarray(gray)
        if gray.dtype != np.uint8 or gray.ndim != 2 or gray.strides[1] != 1:
            raise ValueError("need a 2-D uint8 image with unit column stride")
        out = np.zeros((capacity, 4), np.int32)
        n = C.c_int()
        levels = self.plan_cv24(gray.shape[0], gray.shape[1], scale_factor, min_size, max_size) if want_taps else []
        taps = np.zeros(max(sum(l["nx"] * l["ny"] for l in levels), 1), np.int8)
        _check(lib.hc_detect_multiscale_cv24_taps(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                                  scale_factor, min_neighbors, flags, min_size[0], min_size[1], max_size[0],
                                                  max_size[1], out.ctypes.data, capacity, C.byref(n),
                                                  taps.ctypes.data if want_taps else None, taps.size if want_taps else 0))
        if not want_taps:
            return out[:n.value].copy()
        per, o = [], 0
        for l in levels:
            per.append(taps[o:o + l["nx"] * l["ny"]].reshape(l["ny"], l["nx"]))
            o += l["nx"] * l["ny"]
        return out[:n.value].copy(), per

    def detect_faces(self, bgr, lower=SKIN_LOWER, upper=SKIN_UPPER, scale=1.1, capacity=4096):
        """FaceDetector::detectFaces(frame, rects) (reference src/tracker/detector.cc:10-43)"""
        bgr = np.asarray(bgr)
        if bgr.dtype != np.uint8 or bgr.ndim != 3 or bgr.shape[2] != 3 or bgr.strides[1] != 3 or bgr.strides[2] != 1:
            raise ValueError("need an HxWx3 uint8 BGR image with packed pixels")
        out = np.zeros((capacity, 4), np.int32)
        n = C.c_int()
        _check(lib.hc_detect_faces(self._h, bgr.ctypes.data, bgr.shape[0], bgr.shape[1], bgr.strides[0], _i3(lower),
                                   _i3(upper), scale, out.ctypes.data, capacity, C.byref(n)))
        return out[:n.value].copy()

    def detect_multiscale(self, gray, scale_factor=1.1, min_neighbors=3, flags=0, min_size=(0, 0), max_size=(0, 0),
                          capacity=4096):
        """detectMultiScale(image, scaleFactor, minNeighbors, flags, minSize, maxSize) -> (K,4) int32 x,y,w,h."""
        gray = np.asarray(gray)
        if gray.dtype == np.uint8 and gray.ndim == 3 and gray.shape[2] in (3, 4) and gray.strides[2] == 1 and \
                gray.strides[1] == gray.shape[2]:
            # multi-channel input: upstream converts BGR(A) -> gray inside detectMultiScale
            out = np.zeros((capacity, 4), np.int32)
            n = C.c_int()
            _check(lib.hc_detect_multiscale_image(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.shape[2],
                                                  gray.strides[0], scale_factor, min_neighbors, flags, min_size[0],
                                                  min_size[1], max_size[0], max_size[1], out.ctypes.data, capacity,
                                                  C.byref(n)))
            return out[:n.value].copy()
        if gray.dtype != np.uint8 or gray.ndim != 2 or gray.strides[1] != 1:
            raise ValueError("need a 2-D uint8 image with unit column stride (or HxWx3 / HxWx4 packed BGR / BGRA)")
        out = np.zeros((capacity, 4), np.int32)
        n = C.c_int()
        _check(lib.hc_detect_multiscale(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                        scale_factor, min_neighbors, flags, min_size[0], min_size[1], max_size[0],
                                        max_size[1], out.ctypes.data, capacity, C.byref(n)))
        return out[:n.value].copy()


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


class Context:
    """Batched, device-resident detection context (one stream, one workspace)."""

    def __init__(self, cascade, device=0, stream=None):
        self.cascade = cascade
        self._h = C.c_void_p()
        _check(lib.hc_context_create(cascade._h, device, C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.max_batch = 0

    def stream(self):
        """the hipStream_t (as an int) this context enqueues on: pass it to Context(.., stream=) of the next pipeline slot"""
        return lib.hc_context_stream(self._h)

    def close(self):
        if self._h:
            lib.hc_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, rows, cols, scale_factor=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), max_batch=1,
                  cand_capacity=0):
        _check(lib.hc_context_configure(self._h, rows, cols, scale_factor, min_neighbors, min_size[0], min_size[1],
                                        max_size[0], max_size[1], max_batch, cand_capacity))
        self.rows, self.cols, self.max_batch = rows, cols, max_batch

    def levels(self):
        buf = (Level * 256)()
        n = C.c_int()
        _check(lib.hc_context_levels(self._h, buf, 256, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def algorithmic_bytes(self):
        return int(lib.hc_context_algorithmic_bytes(self._h))

    def enqueue_device(self, dptr, nframes, frame_stride=None, row_step=None):
        _check(lib.hc_detect_batch_enqueue(self._h, C.c_void_p(dptr), frame_stride or self.rows * self.cols,
                                           row_step or self.cols, nframes))

    def set_semantics(self, cv24):
        """False: OpenCV >= 4 (default); True: OpenCV 2.4.x (hc_context_set_semantics); call before configure()"""
        _check(lib.hc_context_set_semantics(self._h, 1 if cv24 else 0))
        self.cv24 = bool(cv24)

    def set_device_grouping(self, on=True):
        """rectangles grouped by the batch's own last kernels (K6) instead of on the host; same boxes, same order"""
        _check(lib.hc_context_set_device_grouping(self._h, int(on)))

    def group_host_frames(self):
        """frames of the last waited batch whose rectangles the HOST grouped (0 when the device grouped them all)"""
        return int(lib.hc_context_group_host_frames(self._h))

    def enqueue_keys(self, keys, nframes):
        """test entry: raw candidate keys (frame << 40 | level << 32 | gy << 16 | gx) in place of a batch's candidates"""
        keys = np.ascontiguousarray(keys, np.uint64)
        _check(lib.hc_debug_enqueue_keys(self._h, keys.ctypes.data, len(keys), nframes))
        self._n = nframes

    def set_rerun_on_overflow(self, allow=True):
        """caller-owned device frames stay valid and unmodified until wait(): a candidate-workspace overflow re-runs by itself"""
        _check(lib.hc_context_set_rerun_on_overflow(self._h, int(allow)))

    def enqueue_host(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.ndim == 3 and frames.shape[1:] == (self.rows, self.cols)
        self._keep = frames
        _check(lib.hc_detect_batch_enqueue_host(self._h, frames.ctypes.data, frames.strides[0], frames.strides[1],
                                                frames.shape[0]))
        self._n = frames.shape[0]

    def wait(self, nframes=None, capacity=1 << 16):
        nframes = nframes if nframes is not None else self._n
        if getattr(self, "_out", None) is None or len(self._out) < capacity:
            self._out = np.empty((capacity, 4), np.int32)
        out = self._out
        ofs = np.zeros(nframes + 1, np.int32)
        _check(lib.hc_detect_batch_wait(self._h, out.ctypes.data, capacity, ofs.ctypes.data))
        allb = out[:ofs[nframes]].copy()
        return [allb[ofs[i]:ofs[i + 1]] for i in range(nframes)]

    def raw_candidates(self, nframes, capacity=1 << 20):
        out = np.zeros((capacity, 4), np.int32)
        ofs = np.zeros(nframes + 1, np.int32)
        _check(lib.hc_batch_raw_candidates(self._h, out.ctypes.data, capacity, ofs.ctypes.data))
        return [out[ofs[i]:ofs[i + 1]].copy() for i in range(nframes)]

    def detect_host(self, frames):
        self.enqueue_host(frames)
        return self.wait(len(frames))

    def enable_faces(self):
        _check(lib.hc_context_enable_faces(self._h))

    def enable_scan_resize(self, src_rows, src_cols):
        _check(lib.hc_context_enable_scan_resize(self._h, src_rows, src_cols))
        self._src = (src_rows, src_cols)

    def faces_host(self, bgr_frames, lower=SKIN_LOWER, upper=SKIN_UPPER, capacity=1 << 14):
        """FaceDetector::detectFaces on a batch of BGR frames [n][rows][cols][3] -> list of (K,4) rects"""
        fr = np.ascontiguousarray(bgr_frames, np.uint8)
        assert fr.ndim == 4 and fr.shape[1:] == getattr(self, "_src", (self.rows, self.cols)) + (3,)
        _check(lib.hc_faces_batch_enqueue_host(self._h, fr.ctypes.data, fr.strides[0], fr.strides[1], fr.shape[0],
                                               _i3(lower), _i3(upper)))
        out = np.zeros((capacity, 4), np.int32)
        ofs = np.zeros(fr.shape[0] + 1, np.int32)
        _check(lib.hc_faces_batch_wait(self._h, out.ctypes.data, capacity, ofs.ctypes.data))
        return [out[ofs[i]:ofs[i + 1]].copy() for i in range(fr.shape[0])]

    def read_face_tap(self, frame, what):
        a = np.zeros((self.rows, self.cols, 3) if what == TAP_FACE_BGR else (self.rows, self.cols), np.uint8)
        _check(lib.hc_context_read_tap(self._h, frame, 0, what, a.ctypes.data, a.nbytes))
        return a

    def enable_stage_taps(self, on=True):
        _check(lib.hc_context_enable_stage_taps(self._h, int(on)))

    def enable_timing(self, on=True):
        _check(lib.hc_context_enable_timing(self._h, int(on)))

    def kernel_ms(self):
        ms = (C.c_float * HC_NKERNELS)()
        _check(lib.hc_context_kernel_ms(self._h, ms))
        return {lib.hc_kernel_name(i).decode(): float(ms[i]) for i in range(HC_NKERNELS)}

    def compiled_stages(self):
        """leading stages that ran as compiled code in every cascade launch of the last enqueued batch (0: generic kernel)"""
        n = C.c_int()
        _check(lib.hc_context_compiled_stages(self._h, C.byref(n)))
        return n.value

    def last_launch(self):
        """what the last enqueue launched, as a dict (hc_context_last_launch): kernel variant per scan-step class
        ('variant': (step 1, step 2) names), prep / variance route names, and the frame-chunk values of both remaps"""
        info = LaunchInfo()
        _check(lib.hc_context_last_launch(self._h, C.byref(info)))
        d = {k: getattr(info, k) for k, _ in LaunchInfo._fields_ if k not in ("variant", "compiled_stages", "band_step", "lds_bytes")}
        d["band_step"] = tuple(info.band_step)
        d["lds_bytes"] = tuple(info.lds_bytes)
        d["variant"] = tuple(VARIANT_NAMES[v] for v in info.variant)
        d["compiled_stages"] = tuple(info.compiled_stages)
        d["prep_route"] = PREP_NAMES[info.prep_route]
        d["variance_route"] = VARIANCE_NAMES[info.variance_route]
        return d

    def read_tap(self, frame, level_index, what):
        L = self.levels()[level_index]
        if what == TAP_PYRAMID:
            a = np.zeros((L.h, L.w), np.uint8)
        elif what in (TAP_SUM, TAP_SQSUM):
            a = np.zeros((L.h + 1, L.w + 1), np.int32)
        elif what == TAP_VNF:
            a = np.zeros((L.ny, L.nx), np.float32)
        else:
            a = np.zeros(((L.szw_h + L.ystep - 1) // L.ystep, L.nx), np.int8)
        if a.size:
            _check(lib.hc_context_read_tap(self._h, frame, level_index, what, a.ctypes.data, a.nbytes))
        return a


def load_spec_plugin(path):
    """register compiled early stages built for another cascade (tools/build_spec_plugin.sh)"""
    _check(lib.hc_spec_plugin_load(os.fsencode(path)))


def alloc_count():
    """device / pinned-host allocations the library has made in this process (hc_debug_alloc_count)"""
    return int(lib.hc_debug_alloc_count())


def configure_shared(contexts, rows, cols, scale_factor=1.1, min_neighbors=3, min_size=(0, 0), max_size=(0, 0), max_batch=1,
                     cand_capacity=0):
    """hc_context_configure_shared: contexts of different cascades over the same frames, one pyramid + integral pass
    (contexts[0] leads: enqueue it first, then every follower once, per batch)."""
    arr = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    _check(lib.hc_context_configure_shared(arr, len(contexts), rows, cols, scale_factor, min_neighbors, min_size[0],
                                           min_size[1], max_size[0], max_size[1], max_batch, cand_capacity))
    for c in contexts:
        c.rows, c.cols, c.max_batch = rows, cols, max_batch


def resize_linear_cv24(img, drows, dcols):
    """cvResize(CV_INTER_LINEAR) of OpenCV 2.4 on the device (hc_resize_linear_cv24)"""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1:
        raise ValueError("need a 2-D uint8 image with unit column stride")
    dst = np.zeros((drows, dcols), np.uint8)
    _check(lib.hc_resize_linear_cv24(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], dst.ctypes.data, drows, dcols))
    return dst


def bgr2gray(img):
    """cvtColor(img, COLOR_BGR2GRAY) on the device (hc_convert_bgr2gray): HxWx3 / HxWx4 uint8 -> HxW uint8"""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] not in (3, 4) or img.strides[2] != 1 or img.strides[1] != img.shape[2]:
        raise ValueError("need an HxWx3 / HxWx4 uint8 image with packed pixels")
    g = np.zeros(img.shape[:2], np.uint8)
    _check(lib.hc_convert_bgr2gray(img.ctypes.data, img.shape[0], img.shape[1], img.shape[2], img.strides[0], g.ctypes.data,
                                   g.strides[0]))
    return g


def save_boxes(path, boxes_by_frame):
    """HCBX box file through the C ABI (hc_boxes_save): dict frame_id -> (K,4) int array"""
    ids = np.array(list(boxes_by_frame), np.int32)
    lists = [np.asarray(boxes_by_frame[int(i)], np.int32).reshape(-1, 4) for i in ids]
    ofs = np.zeros(len(ids) + 1, np.int32)
    ofs[1:] = np.cumsum([len(b) for b in lists])
    rects = np.ascontiguousarray(np.concatenate(lists) if lists else np.zeros((0, 4), np.int32))
    _check(lib.hc_boxes_save(os.fsencode(path), len(ids), ids.ctypes.data, ofs.ctypes.data, rects.ctypes.data))


def load_boxes(path):
    """HCBX box file through the C ABI (hc_boxes_load) -> dict frame_id -> (K,4) int32, ascending frame id"""
    nf, nr = C.c_int(), C.c_int()
    _check(lib.hc_boxes_load(os.fsencode(path), None, None, 0, None, 0, C.byref(nf), C.byref(nr)))     # size query
    ids = np.zeros(nf.value, np.int32)
    ofs = np.zeros(nf.value + 1, np.int32)
    rects = np.zeros((max(nr.value, 1), 4), np.int32)
    _check(lib.hc_boxes_load(os.fsencode(path), ids.ctypes.data, ofs.ctypes.data, nf.value, rects.ctypes.data, nr.value,
                             C.byref(nf), C.byref(nr)))
    return {int(ids[i]): rects[ofs[i]:ofs[i + 1]].copy() for i in range(nf.value)}


def group_rectangles(rects, group_threshold, eps=0.2):
    rects = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(-1, 4))
    out = np.zeros((max(len(rects), 1), 4), np.int32)
    n = C.c_int()
    _check(lib.hc_group_rectangles(rects.ctypes.data, len(rects), group_threshold, eps, out.ctypes.data, len(out),
                                   C.byref(n)))
    return out[:n.value].copy()
