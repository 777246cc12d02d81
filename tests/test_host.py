"""CPU tests of the PRODUCT's host logic and of the C-ABI surface (no GPU, no compute calls):
the shared library loads and exports every symbol declared in include/hip_cascade.h, the
cascade loader / scale plan / rectangle grouping agree with the oracle, and the error
behaviour mirrors the reference's conventions (int status codes, load -> false)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "hip_cascade.h")).read()
    declared = sorted(set(re.findall(r"HC_API\s+[\w\s\*]+?\b(hc_\w+)\s*\(", hdr)))
    assert len(declared) >= 24
    lib = C.CDLL(pkg.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libhip_cascade.so does not export %s" % name
    assert declared == pkg.EXPORTED, "python binding and header disagree: %s" % (set(declared) ^ set(pkg.EXPORTED))
    # plain C ABI: no C++ / torch types may appear in the header
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    assert "std::" not in code and "torch" not in code and "at::" not in code and "class " not in code
    assert pkg.lib.hc_version().decode().startswith("hip_cascade")
    assert [pkg.lib.hc_kernel_name(i).decode() for i in range(pkg.HC_NKERNELS)] == [
        "k_pyramid_integral", "k_integral_rows", "k_integral_cols", "k_variance", "k_cascade"]


@pytest.mark.parametrize("name", ["alt", "alt2", "alt_tree", "default"])
def test_loader_matches_oracle(pkg, oracle, name):
    a = pkg.Cascade(oracle.cascade_path(name))
    o = oracle.OracleCascade(oracle.cascade_path(name))
    assert (a.win_w, a.win_h, a.nstages, a.ntrees, a.nnodes, a.nleaves) == (o.win_w, o.win_h, o.nstages, o.ntrees,
                                                                          o.nnodes, o.nleaves)
    assert a.is_tree == (1 if o.max_nodes > 1 else 0)
    for x, y in zip(a.dump(), o.dump()):
        assert x.tobytes() == y.tobytes()
    xml = oracle.cascade_path(name, xml=True)
    if os.path.exists(xml):      # the reference's own file, parsed by the product's XML scanner
        for x, y in zip(pkg.Cascade(xml).dump(), o.dump()):
            assert x.tobytes() == y.tobytes()


def test_loader_errors(pkg, tmp_path):
    with pytest.raises(pkg.HipCascadeError) as e:
        pkg.Cascade(str(tmp_path / "missing.xml"))
    assert e.value.code == -1                                   # HC_ERR_IO
    bad = tmp_path / "bad.xml"
    bad.write_text("<?xml version='1.0'?><opencv_storage><x type_id='opencv-matrix'>1</x></opencv_storage>")
    with pytest.raises(pkg.HipCascadeError) as e:
        pkg.Cascade(str(bad))
    assert e.value.code == -2                                   # HC_ERR_FORMAT
    trunc = tmp_path / "trunc.hcc"
    trunc.write_bytes(b"HCC1" + b"\x00" * 10)
    with pytest.raises(pkg.HipCascadeError) as e:
        pkg.Cascade(str(trunc))
    assert e.value.code == -2


def test_minimal_xml_roundtrip(pkg, oracle, tmp_path):
    """A hand-written 1-stage old-format file: both loaders read the same numbers (incl. '-1.' and
    exponent forms, comments, a tree with an internal node)."""
    xml = tmp_path / "tiny.xml"
    xml.write_text("""<?xml version="1.0"?>
<!-- comment -->
<opencv_storage>
<tiny type_id="opencv-haar-classifier">
  <size>20 20</size>
  <stages>
    <_>
      <!-- stage 0 -->
      <trees>
        <_>
          <_>
            <feature><rects><_>3 7 14 4 -1.</_><_>3 9 14 2 2.</_></rects><tilted>0</tilted></feature>
            <threshold>4.0141958743333817e-003</threshold>
            <left_val>0.0337941907346249</left_val>
            <right_node>1</right_node></_>
          <_>
            <feature><rects><_>1 2 18 4 -1.</_><_>7 2 6 4 3.</_></rects><tilted>0</tilted></feature>
            <threshold>-1.5e-002</threshold>
            <left_val>-0.25</left_val>
            <right_val>0.75</right_val></_></_></trees>
      <stage_threshold>0.5</stage_threshold>
      <parent>-1</parent>
      <next>-1</next></_></stages></tiny>
</opencv_storage>
""")
    a = pkg.Cascade(str(xml))
    o = oracle.OracleCascade(str(xml))
    assert (a.nstages, a.ntrees, a.nnodes, a.nleaves, a.is_tree) == (1, 1, 2, 3, 1)
    for x, y in zip(a.dump(), o.dump()):
        assert x.tobytes() == y.tobytes()
    st, sn, tn, nr, nt, nl, nrr, lf = a.dump()
    assert st[0] == np.float32(np.float32(0.5) - np.float32(1e-5))
    assert nl.tolist() == [0, -1] and nrr.tolist() == [1, -2]
    assert lf.tolist() == [np.float32(0.0337941907346249), -0.25, 0.75]


def test_plan_matches_oracle(pkg, oracle):
    rng = np.random.default_rng(5)
    for name in ("alt", "default"):
        a = pkg.Cascade(oracle.cascade_path(name))
        o = oracle.OracleCascade(oracle.cascade_path(name))
        cases = [(720, 1280, 1.1, (30, 30), (0, 0)), (360, 640, 1.1, (30, 30), (0, 0)), (1080, 1920, 1.1, (30, 30), (0, 0)),
                 (2160, 3840, 1.1, (30, 30), (0, 0)), (19, 19, 1.1, (0, 0), (0, 0)), (24, 24, 1.1, (0, 0), (0, 0)),
                 (200, 200, 1.1, (77, 77), (77, 77)), (300, 200, 1.25, (0, 0), (120, 120))]
        for _ in range(40):
            cases.append((int(rng.integers(10, 500)), int(rng.integers(10, 700)), float(rng.choice([1.05, 1.1, 1.2, 1.5])),
                          (int(rng.integers(0, 60)),) * 2, (0, 0)))
        for rows, cols, sf, mn, mx in cases:
            la, ab = a.plan(rows, cols, sf, mn, mx)
            lo, _ = o.plan(rows, cols, sf, mn, mx)
            assert len(la) == len(lo), (rows, cols, sf, mn, mx)
            for x, y in zip(la, lo):
                assert (x.w, x.h, x.ystep, x.szw_w, x.szw_h, x.y_end, x.nx, x.ny, x.win_w, x.win_h) == (
                    y.w, y.h, y.ystep, y.szw_w, y.szw_h, y.y_end, y.nx, y.ny, y.win_w, y.win_h)
                assert np.float32(x.scale).tobytes() == np.float32(y.scale).tobytes()
            assert ab == rows * cols + sum(16 * (l.w + 1) * (l.h + 1) for l in lo)
    a = pkg.Cascade(oracle.cascade_path("alt"))
    assert a.plan(720, 1280, 1.1, (30, 30))[1] == 33813536
    assert a.plan(360, 640, 1.1, (30, 30))[1] == 8464320
    with pytest.raises(pkg.HipCascadeError) as e:
        a.plan(100, 100, 1.0)
    assert e.value.code == -3                                    # scaleFactor <= 1 -> HC_ERR_ARG


def test_group_rectangles_matches_oracle(pkg, oracle):
    rng = np.random.default_rng(6)
    assert pkg.group_rectangles(np.zeros((0, 4), np.int32), 2).tolist() == []
    for it in range(200):
        n = int(rng.integers(1, 120))
        centers = rng.integers(0, 400, (6, 2))
        rects = []
        for _ in range(n):
            cx, cy = centers[rng.integers(0, 6)]
            s = int(rng.choice([30, 33, 36, 40, 44, 80, 88]))
            rects.append([int(cx + rng.integers(-8, 9)), int(cy + rng.integers(-8, 9)), s, s + int(rng.integers(0, 2))])
        thr = int(rng.integers(0, 4))
        assert pkg.group_rectangles(rects, thr).tolist() == oracle.group_rectangles(rects, thr).tolist(), it


def test_compute_entry_points_fail_loudly_without_device(pkg, oracle):
    """No CPU fallback: on a machine without a HIP device every compute call reports
    HC_ERR_NO_DEVICE instead of quietly computing something else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    c = pkg.Cascade(oracle.cascade_path("alt"))
    with pytest.raises(pkg.HipCascadeError) as e:
        c.detect_multiscale(np.zeros((64, 64), np.uint8))
    assert e.value.code == -5
    with pytest.raises(pkg.HipCascadeError) as e:
        pkg.Context(c)
    assert e.value.code == -5


def test_cpp_shim_builds_and_reports_errors(pkg, oracle, tmp_path):
    """hip_cascade::CascadeClassifier (the cv::CascadeClassifier stand-in) through its demo binary:
    load() failure -> exit 1 (reference convention src/file_io.cc:11-13)."""
    demo = os.path.join(ROOT, "ugproj-detector_amd", "shim_demo")
    assert os.path.exists(demo), "shim_demo not built (make -C ugproj-detector_amd)"
    fr = tmp_path / "f.u8"
    np.zeros((64, 64), np.uint8).tofile(fr)
    r = subprocess.run([demo, str(tmp_path / "nope.xml"), str(fr), "64", "64"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot load cascade" in r.stderr


def test_box_file_through_the_c_abi(pkg, tmp_path):
    """SURVEY 8f-4 for the C++ side: hc_boxes_save / hc_boxes_load write and read the very bytes of
    ugproj-detector_amd/shard.py's HCBX format, so the C++ tracker can consume what a Python / multi-GPU run wrote
    and vice versa (reference src/tracker/face_tracker.cc:294-302 consumes every box of every frame in order)."""
    import importlib
    shard = importlib.import_module("ugproj_detector_amd.shard")
    boxes = {7: np.array([[1, 2, 30, 30]], np.int32), 3: np.zeros((0, 4), np.int32),
             11: np.arange(4 * 100, dtype=np.int32).reshape(100, 4), 2_000_000: np.array([[-5, 0, 9, 9], [1, 1, 1, 1]], np.int32)}
    a, b = str(tmp_path / "c.hcbx"), str(tmp_path / "py.hcbx")
    pkg.save_boxes(a, boxes)                 # C ABI writer (frames handed over in dict order, stored in ascending id)
    shard.save_boxes(b, boxes)               # Python writer
    assert open(a, "rb").read() == open(b, "rb").read()
    for path in (a, b):
        got = pkg.load_boxes(path)           # C ABI reader
        assert list(got) == sorted(boxes)
        for k in boxes:
            assert got[k].tolist() == boxes[k].tolist()
    assert {k: v.tolist() for k, v in shard.load_boxes(a).items()} == {k: v.tolist() for k, v in boxes.items()}
    # empty file, size query, too-small buffers, malformed input, duplicate ids
    e = str(tmp_path / "empty.hcbx")
    pkg.save_boxes(e, {})
    assert pkg.load_boxes(e) == {}
    import ctypes as C
    nf, nr = C.c_int(), C.c_int()
    assert pkg.lib.hc_boxes_load(a.encode(), None, None, 0, None, 0, C.byref(nf), C.byref(nr)) == 0
    assert (nf.value, nr.value) == (4, 103)
    ids = np.zeros(2, np.int32)
    ofs = np.zeros(3, np.int32)
    rects = np.zeros((103, 4), np.int32)
    rc = pkg.lib.hc_boxes_load(a.encode(), ids.ctypes.data, ofs.ctypes.data, 2, rects.ctypes.data, 103, C.byref(nf), C.byref(nr))
    assert rc == -7 and (nf.value, nr.value) == (4, 103)               # HC_ERR_CAPACITY, sizes still reported
    raw = open(a, "rb").read()
    for bad in (b"nope", raw[:10], raw[:-3], raw[:12]):
        (tmp_path / "bad").write_bytes(bad)
        rc = pkg.lib.hc_boxes_load(str(tmp_path / "bad").encode(), None, None, 0, None, 0, C.byref(nf), C.byref(nr))
        assert rc == -2, bad[:12]                                       # HC_ERR_FORMAT
    assert pkg.lib.hc_boxes_load(str(tmp_path / "missing").encode(), None, None, 0, None, 0, C.byref(nf), C.byref(nr)) == -1
    dup_ids = np.array([5, 5], np.int32)
    dofs = np.array([0, 0, 0], np.int32)
    assert pkg.lib.hc_boxes_save(str(tmp_path / "dup").encode(), 2, dup_ids.ctypes.data, dofs.ctypes.data, None) == -3


def test_cv24_plan_matches_oracle_without_a_device(pkg, oracle):
    """hc_plan_query_cv24 (pure host): the legacy mode's f64 ladder and scan grids equal the oracle's for several
    geometries / scale factors / size limits and both window sizes."""
    for name in ("alt", "default"):
        casc, oc = pkg.Cascade(oracle.cascade_path(name)), oracle.OracleCascade(oracle.cascade_path(name))
        for rows, cols, sf, mins, maxs in [(720, 1280, 1.1, (30, 30), (0, 0)), (360, 640, 1.1, (30, 30), (0, 0)),
                                           (97, 131, 1.2, (0, 0), (0, 0)), (240, 320, 1.05, (40, 40), (90, 90)), (19, 40, 1.1, (0, 0), (0, 0)),
                                           (24, 24, 1.3, (0, 0), (0, 0))]:
            a, b = casc.plan_cv24(rows, cols, sf, mins, maxs), oc.plan_cv24(rows, cols, sf, mins, maxs)
            assert a == b, (name, rows, cols, sf)
    import ctypes as C
    n = C.c_int()
    assert pkg.lib.hc_detect_multiscale_cv24(None, None, 1, 1, 1, 1.1, 2, 2, 0, 0, 0, 0, None, 0, C.byref(n)) == -3


def test_shim_compiles_with_opencv_spelling(tmp_path):
    """The `HIP_CASCADE_HAVE_OPENCV` branch of the C++ shim (cv::Mat / cv::Rect / cv::Size aliases: what the reference's
    tracker gets) had never been through a compiler: no OpenCV exists in this image.  Syntax
    check against tests/mock_opencv (a labelled mock of the few declarations the shim touches, not OpenCV), driving
    the reference's two calls -- src/file_io.cc:11 and src/tracker/detector.cc:25-31 -- with cv:: types."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_shim.cpp"
    src.write_text("""
#include <opencv2/core/core.hpp>
#include "hip_cascade/cascade_classifier.hpp"
#include "hip_cascade/face_detector.hpp"
#ifndef HIP_CASCADE_HAVE_OPENCV
#error "the OpenCV branch of the shim was not selected"
#endif
using namespace cv;
int run(const Mat& gray, const Mat& frame) {
  hip_cascade::CascadeClassifier cascade;                       // src/file_io.h:38
  if (!cascade.load("haarcascade_frontalface_alt.xml")) return 1;   // src/file_io.cc:11
  std::vector<Rect> facesInGray;
  cascade.detectMultiScale(gray, facesInGray, 1.1, 2, 0 | hip_cascade::CASCADE_SCALE_IMAGE, Size(30, 30));   // detector.cc:25-31
  hip_cascade::DetectionSection cfg;
  hip_cascade::FaceDetector detector(cascade, cfg);             // src/tracker/face_tracker.cc:41-43
  std::vector<Rect> rects;
  detector.detectFaces(frame, rects);                           // src/tracker/face_tracker.cc:295
  static_assert(std::is_same<hip_cascade::Rect, cv::Rect>::value && std::is_same<hip_cascade::Mat, cv::Mat>::value, "aliases");
  return (int)(facesInGray.size() + rects.size());
}
""")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-I", os.path.join(root, "tests", "mock_opencv"),
                        "-I", os.path.join(root, "ugproj-detector_amd"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_compiled_stages_belong_to_exactly_their_cascade(pkg, oracle, tmp_path):
    """The library carries generated straight-line code for the early stages of the four stock cascades (csrc/spec/,
    tools/gen_spec.py).  It may be used only for a cascade that is bit for bit the one it was generated from: the
    match is a hash over every number that decides a result.  Host-only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = {}
    for name, want in (("alt", True), ("default", True), ("alt_tree", True), ("alt2", True)):
        h, n = pkg.Cascade(oracle.cascade_path(name)).compiled_stages()
        hashes[name] = h
        assert (n > 0) == want, (name, n)
    assert len(set(hashes.values())) == 4
    # one leaf value changed in the last stage -> different hash, generic kernel
    raw = bytearray(open(oracle.cascade_path("alt"), "rb").read())
    raw[-4:] = np.float32(np.frombuffer(bytes(raw[-4:]), np.float32)[0] * np.float32(1.5)).tobytes()
    mod = tmp_path / "alt_mod.hcc"
    mod.write_bytes(bytes(raw))
    h, n = pkg.Cascade(str(mod)).compiled_stages()
    assert n == 0 and h != hashes["alt"]
    # the committed generated files are what the generator writes today, and carry the hash the library computes
    for tag, name in (("alt", "alt"), ("dflt", "default"), ("atree", "alt_tree"), ("alt2", "alt2")):
        inc = os.path.join(root, "ugproj-detector_amd", "csrc", "spec", "spec_%s.inc" % tag)
        text = open(inc).read()
        assert "0x%016xull" % hashes[name] in text
        nst = int(re.search(r"nstages = (\d+)", text).group(1))
        out = tmp_path / ("spec_%s.inc" % tag)
        subprocess.check_call(["python3", os.path.join(root, "tools", "gen_spec.py"), oracle.cascade_path(name), tag, str(nst), str(out)],
                              stdout=subprocess.DEVNULL)
        assert out.read_text() == text, "csrc/spec/spec_%s.inc is stale: run make -C ugproj-detector_amd specs" % tag
    # the units of the 16-bit-tile build (scan step 1) come from the same generator with --p16: alt; default, whose 24 x 24 window
    # has rects of up to 576 pixels in its compiled stages (cut in two, complemented by rect 1's side strips, or -- checkerboard
    # stumps -- replaced by their quadrants, so that every rect sum formed is of at most 257 pixels); alt_tree
    for name, tag in (("alt", "altp"), ("default", "dfltp"), ("alt_tree", "atreep")):
        inc = os.path.join(root, "ugproj-detector_amd", "csrc", "spec", "spec_%s.inc" % tag)
        text = open(inc).read()
        assert "0x%016xull" % hashes[name] in text and "p16 = true" in text
        out = tmp_path / ("spec_%s.inc" % tag)
        subprocess.check_call(["python3", os.path.join(root, "tools", "gen_spec.py"), oracle.cascade_path(name), tag,
                               re.search(r"nstages = (\d+)", text).group(1), str(out), "--p16"], stdout=subprocess.DEVNULL)
        assert out.read_text() == text, "csrc/spec/spec_%s.inc is stale: run make -C ugproj-detector_amd specs" % tag
    # ... and the units with OpenCV 2.4's node arithmetic with --cv24 (hidden f32 rect weights as immediates)
    for name, tag in (("alt", "alt24"), ("alt2", "alt224"), ("default", "dflt24"), ("alt_tree", "atree24")):
        inc = os.path.join(root, "ugproj-detector_amd", "csrc", "spec", "spec_%s.inc" % tag)
        text = open(inc).read()
        assert "0x%016xull" % hashes[name] in text and "cv24 = true" in text
        out = tmp_path / ("spec_%s.inc" % tag)
        subprocess.check_call(["python3", os.path.join(root, "tools", "gen_spec.py"), oracle.cascade_path(name), tag,
                               re.search(r"nstages = (\d+)", text).group(1), str(out), "--cv24"], stdout=subprocess.DEVNULL)
        assert out.read_text() == text, "csrc/spec/spec_%s.inc is stale: run make -C ugproj-detector_amd specs" % tag
    # a tree cascade cannot take that build
    r = subprocess.run(["python3", os.path.join(root, "tools", "gen_spec.py"), oracle.cascade_path("alt2"), "x", "6", str(tmp_path / "x.inc"),
                        "--p16"], capture_output=True, text=True)
    assert r.returncode != 0 and "stump" in (r.stderr + r.stdout)


@pytest.mark.parametrize("name,shape,comp", [("alt", (720, 1280), 0), ("default", (720, 1280), 0), ("alt2", (1080, 1920), 0),
                                             ("default", (2160, 3840), 0), ("alt", (240, 320), 96), ("alt_tree", (203, 331), 64),
                                             ("alt", (97, 131), 0), ("default", (1080, 1920), 256)])
def test_strip_plan_invariants(pkg, oracle, name, shape, comp):
    """The planner of the fused prep kernel's column strips (pure host code): the strips of a level tile its columns exactly and
    left to right, every strip computes its own windows' halo, owns at most one register ring per thread, stages at most 2 KB of
    a source row, and the whole workgroup fits two to a CU; the hand-off links and carry blocks are consistent."""
    casc = pkg.Cascade(oracle.cascade_path(name))
    levels, _ = casc.plan(shape[0], shape[1], 1.1, (30, 30))
    strips, lds = casc.plan_strips(shape[0], shape[1], 1.1, (30, 30), comp_max=comp)
    assert len(strips) >= len(levels) and 0 < lds <= 81920
    nw = casc.win_w - 2
    by_level = {}
    for i, (lv, c0, c1, ce, g0, g1, sx0, sxn, left, has_right, carry_ofs, _z) in enumerate(strips.tolist()):
        by_level.setdefault(lv, []).append(i)
        L = levels[lv]
        assert 0 <= c0 < c1 <= ce <= L.w
        assert ce - c0 <= (comp if comp else 832) and g1 - g0 <= 512 and sxn <= 2048 and sx0 % 4 == 0 and sxn % 4 == 0
        assert 0 <= g0 <= g1 <= L.nx
        for g in range(g0, g1):                                # a window's inner-rect box sum lies inside the computed columns
            assert c0 <= g * L.ystep and g * L.ystep + nw < ce
        # the staged span holds both bilinear taps of every computed pixel (source x of dest x: (x + 0.5) cols / w - 0.5; one
        # column of slack for the fixed-point coefficient rule)
        first_tap = max(0, int(np.floor((c0 + 0.5) * shape[1] / L.w - 0.5)))
        last_tap = min(shape[1] - 1, int(np.floor((ce - 1 + 0.5) * shape[1] / L.w - 0.5)) + 1)
        assert sx0 <= first_tap + 1 and sx0 + sxn >= last_tap, (lv, c0, ce, sx0, sxn, first_tap, last_tap)
    assert sorted(by_level) == list(range(len(levels)))        # widest levels first, every level present
    carry_blocks = []
    for lv, idx in by_level.items():
        L = levels[lv]
        assert idx == list(range(idx[0], idx[0] + len(idx)))   # strips of a level are contiguous, left to right
        rows_ = strips[idx]
        assert rows_[0, 1] == 0 and rows_[-1, 2] == L.w and all(rows_[k, 2] == rows_[k + 1, 1] for k in range(len(idx) - 1))
        assert rows_[0, 4] == 0 and rows_[-1, 5] == L.nx and all(rows_[k, 5] == rows_[k + 1, 4] for k in range(len(idx) - 1))
        for k, i in enumerate(idx):
            assert strips[i, 8] == (idx[k - 1] if k else -1)    # left neighbour: the previous strip of the level (smaller index)
            assert strips[i, 9] == (1 if k + 1 < len(idx) else 0)
            if strips[i, 9]:
                carry_blocks.append((int(strips[i, 10]), 2 * L.h))
    carry_blocks.sort()
    for (a, n), (b, _) in zip(carry_blocks, carry_blocks[1:]):
        assert a % 2 == 0 and a + n <= b                        # 8-byte aligned, disjoint
    if comp == 0 and shape == (720, 1280) and name == "alt":
        assert sum(len(v) > 1 for v in by_level.values()) == 2  # only the two scan-step-1 levels with more than 512 grid columns are cut


def test_cpp_synthetic_frames_equal_the_python_generator(pkg, synth, tmp_path):
    """SURVEY 8d asks for ONE deterministic generator in C++ and Python: hip_cascade/synth.hpp (what throughput_demo and
    multi_gpu_demo feed themselves with) writes the same bytes as synth.py for every family, ragged sizes included."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "synth_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-I", os.path.join(root, "ugproj-detector_amd", "hip_cascade"),
                           "-o", exe, os.path.join(root, "tests", "tools", "synth_check.cpp")])
    face = os.path.join(root, "ugproj-detector_amd", "data", "face64.u8")
    for kind, rows, cols, first, count in [("noise", 37, 53, 0, 2), ("texture", 61, 83, 5, 2), ("planted", 240, 320, 0, 3),
                                           ("planted", 360, 640, 100, 1), ("planted", 130, 131, 7, 1)]:
        out = str(tmp_path / "f.u8")
        subprocess.check_call([exe, face, kind, str(rows), str(cols), str(first), str(count), out])
        got = np.fromfile(out, np.uint8).reshape(count, rows, cols)
        want = synth.make_frames(kind, count, rows, cols, first=first)
        assert np.array_equal(got, want), (kind, rows, cols, int((got != want).sum()))


def _test_plugin(pkg):
    d = os.path.join(os.path.dirname(pkg.LIB_PATH), "build", "test_plugin")
    so, hcc = os.path.join(d, "libhc_spec_alt2mod.so"), os.path.join(d, "alt2_mod.hcc")
    if not (os.path.exists(so) and os.path.exists(hcc)):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(pkg.LIB_PATH), "test-plugin"])
    return so, hcc


def test_compiled_stage_plugin_registers_for_exactly_its_cascade(pkg, oracle):
    """A cascade the library was not built with gets its compiled early stages from a plug-in
    (tools/build_spec_plugin.sh -> hc_spec_plugin_load): afterwards exactly that cascade reports them.  Host-only (the
    plug-in's kernels are cross-compiled for gfx950; tests/test_gpu_parity.py runs them)."""
    so, hcc = _test_plugin(pkg)
    stock = pkg.Cascade(oracle.cascade_path("alt2")).compiled_stages()
    mod = pkg.Cascade(hcc)
    h0, _ = mod.compiled_stages()
    assert h0 != stock[0]
    pkg.load_spec_plugin(so)
    pkg.load_spec_plugin(so)                                  # registering twice is harmless
    assert mod.compiled_stages() == (h0, 8)
    assert pkg.Cascade(oracle.cascade_path("alt2")).compiled_stages() == stock
    with pytest.raises(pkg.HipCascadeError) as e:
        pkg.load_spec_plugin(pkg.LIB_PATH)                    # a shared object that is not a plug-in
    assert e.value.code == -2
    with pytest.raises(pkg.HipCascadeError):
        pkg.load_spec_plugin("/nonexistent/plugin.so")
