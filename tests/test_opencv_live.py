"""Opportunistic live check against a REAL OpenCV.  OpenCV is absent from the build
container and from the GPU box image, so this module normally reports SKIPPED -- it is never faked.
Wherever `import cv2` works (a developer machine, a future image), it pins the CPU oracle to
cv2.CascadeClassifier.detectMultiScale and to the primitives the path is made of, which would lift
the "parity unpinned" status of DESIGN.md section 2."""
import os

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed here: parity with OpenCV stays unpinned")


def _xml(name):
    ref = "/root/reference/data/haarcascade_frontalface_%s.xml" % name
    if os.path.exists(ref):
        return ref
    p = os.path.join(cv2.data.haarcascades, "haarcascade_frontalface_%s.xml" % name)
    if not os.path.exists(p):
        pytest.skip("no cascade XML reachable for cv2")
    return p


@pytest.mark.parametrize("name", ["alt", "alt2", "default"])
def test_boxes_match_opencv(oracle, synth, name):
    cv2.setNumThreads(1)                       # canonical candidate order: scale -> y -> x
    cc = cv2.CascadeClassifier(_xml(name))
    oc = oracle.OracleCascade(oracle.cascade_path(name))
    for i in range(3):
        fr = synth.planted(240, 320, synth.frame_seed(i), nfaces=3, min_size=40, max_size=110)[0]
        got = cc.detectMultiScale(fr, 1.1, 2, 0 | cv2.CASCADE_SCALE_IMAGE, (30, 30))
        want, raw = oc.detect(fr, 1.1, 2, (30, 30))
        assert oracle.sorted_boxes(np.asarray(got).reshape(-1, 4)).tolist() == oracle.sorted_boxes(want).tolist()
        got0 = cc.detectMultiScale(fr, 1.1, 0, 0, (30, 30))            # minNeighbors 0: raw candidates
        assert oracle.sorted_boxes(np.asarray(got0).reshape(-1, 4)).tolist() == oracle.sorted_boxes(raw).tolist()


def test_primitives_match_opencv(oracle, synth):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (360, 640), dtype=np.uint8)
    for (h, w) in [(224, 397), (168, 299), (21, 38), (180, 320)]:
        assert np.array_equal(cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR_EXACT),
                              oracle.resize_linear_exact(img, h, w))
    s, q = cv2.integral2(img, sdepth=cv2.CV_32S, sqdepth=cv2.CV_32S)
    os_, oq = oracle.integral(img)
    assert np.array_equal(s, os_) and np.array_equal(q.view(np.int32), oq)
    rects = [[10, 10, 30, 30]] * 3 + [[12, 11, 30, 30]] * 2 + [[100, 100, 60, 60]] * 4 + [[110, 110, 20, 20]] * 3
    got, _w = cv2.groupRectangles([list(r) for r in rects], 2, 0.2)
    assert np.asarray(got).reshape(-1, 4).tolist() == oracle.group_rectangles(rects, 2).tolist()
    bgr = synth.planted_bgr(120, 160, synth.frame_seed(3), nfaces=2, min_size=40, max_size=90)[0]
    oc = oracle.OracleCascade(oracle.cascade_path("alt"))
    _r, mask, gray = oc.detect_faces(bgr, want_taps=True)
    ycc = cv2.cvtColor(bgr, cv2.COLOR_BGR2YCrCb)
    m = cv2.inRange(ycc, (0, 133, 77), (255, 173, 127))
    g = cv2.equalizeHist(cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)) & m
    assert np.array_equal(mask, m) and np.array_equal(gray, g)
    big = synth.planted_bgr(240, 320, synth.frame_seed(4), nfaces=2, min_size=60, max_size=120)[0]
    assert np.array_equal(cv2.resize(big, (160, 120), interpolation=cv2.INTER_AREA), oracle.resize_area(big, 120, 160))
