"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle, bit for bit,
at every tap of the path -- pyramid bytes, integral words, variance factors, per-window
stage-pass counts, raw candidates in canonical order, grouped boxes.

Integer/byte/index work: exact equality.  The only floating-point tap (vnf, f32) is also
compared for exact bit equality (tolerance 0): the path is defined by its IEEE operation order.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["policy", "fused", "wholeband"])
def prep_route(request, monkeypatch):
    """Every test of this module runs three times: under the library's own batch-size policy (no environment knob:
    small batches take the three parallel prep kernels, band segments and the side stream; large ones the fused
    k_pyramid_integral and whole bands), with the fused prep kernel pinned (the throughput route, which the
    mostly single-frame batches here would otherwise never reach), and with whole-band cascade launches pinned
    as well -- the instantiations a full batch runs, among them the 16-bit-tile build (HC_P16) that
    haarcascade_frontalface_alt takes for its scan-step-1 levels.  Tests that set a knob themselves override the pin."""
    for k in ("HC_FUSED_PREP", "HC_SEG_MAX_BATCH", "HC_SIDE_MAX_BATCH"):
        monkeypatch.delenv(k, raising=False)
    if request.param != "policy":
        monkeypatch.setenv("HC_FUSED_PREP", "1")
    if request.param == "wholeband":
        monkeypatch.setenv("HC_SEG_MAX_BATCH", "0")
        monkeypatch.setenv("HC_SIDE_MAX_BATCH", "0")
    return request.param


CALL = dict(scale_factor=1.1, min_neighbors=2, min_size=(30, 30))      # reference detector.cc:25-31


def _oracle_taps_per_level(oc, frame, sf, min_size):
    grouped, raw, taps = oc.detect(frame, sf, 2, min_size, (0, 0), want_taps=True)
    return grouped, raw, taps


def _check_frame_all_taps(pkg, oracle, ctx, oc, frame, fidx, sf, min_size):
    levels = ctx.levels()
    olevels, _ = oc.plan(frame.shape[0], frame.shape[1], sf, min_size)
    assert len(levels) == len(olevels)
    grouped, raw, otaps = _oracle_taps_per_level(oc, frame, sf, min_size)
    for li, (L, OL) in enumerate(zip(levels, olevels)):
        assert (L.w, L.h, L.ystep, L.nx, L.ny, L.win_w, L.win_h) == (OL.w, OL.h, OL.ystep, OL.nx, OL.ny, OL.win_w, OL.win_h)
        assert np.float32(L.scale).tobytes() == np.float32(OL.scale).tobytes()
        img = oracle.resize_linear_exact(frame, L.h, L.w)
        pyr = ctx.read_tap(fidx, li, pkg.TAP_PYRAMID)
        assert np.array_equal(pyr, img), "pyramid bytes differ at level %d" % li
        s, q = oracle.integral(img)
        assert np.array_equal(ctx.read_tap(fidx, li, pkg.TAP_SUM), s), "sum differs at level %d" % li
        assert np.array_equal(ctx.read_tap(fidx, li, pkg.TAP_SQSUM), q), "sqsum differs at level %d" % li
        st = ctx.read_tap(fidx, li, pkg.TAP_STAGES)
        assert st.shape == otaps[li].shape
        assert np.array_equal(st, otaps[li]), "stage-pass counts differ at level %d: %d positions" % (
            li, int((st != otaps[li]).sum()))
    return grouped, raw


@pytest.mark.parametrize("name", ["alt", "alt2", "default", "alt_tree"])
def test_all_taps_small_planted(pkg, oracle, synth, name):
    """Every intermediate of the path on a 240x320 planted frame, all four reference cascades."""
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted(240, 320, synth.frame_seed(i), nfaces=3, min_size=40, max_size=110)[0]
                       for i in range(2)])
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=2, cand_capacity=1 << 14)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frames)
    raws = ctx.raw_candidates(2)
    for f in range(2):
        grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, frames[f], f, 1.1, (30, 30))
        assert np.array_equal(raws[f], raw), "raw candidates (canonical order) differ"
        assert np.array_equal(got[f], grouped), "grouped boxes differ (order included)"
    assert sum(len(g) for g in got) > 0


def _vnf_want(oracle, casc, frame, L):
    """A.6a in numpy f64 from the oracle's integrals: the f32 variance factors of every scanned window of a level"""
    img = oracle.resize_linear_exact(frame, L.h, L.w)
    s, q = oracle.integral(img)
    s = s.astype(np.int64)
    q = q.view(np.uint32).astype(np.int64)
    ys = np.arange(L.ny) * L.ystep + 1
    xs = np.arange(L.nx) * L.ystep + 1
    nw, nh = casc.win_w - 2, casc.win_h - 2

    def rs(I, mod):
        v = I[np.ix_(ys, xs)] - I[np.ix_(ys, xs + nw)] - I[np.ix_(ys + nh, xs)] + I[np.ix_(ys + nh, xs + nw)]
        return v % mod if mod else v
    valsum = rs(s, 0).astype(np.float64)
    valsq = rs(q, 1 << 32).astype(np.float64)
    area = float(nw * nh)
    nf = area * valsq - valsum * valsum
    want = np.zeros(nf.shape, np.float32)
    pos = nf > 0
    v = (1.0 / np.sqrt(nf[pos])).astype(np.float32)
    v[~(area * v.astype(np.float64) < 1e-1)] = 0
    want[pos] = v
    return want


def test_vnf_bits(pkg, oracle, synth):
    """A.6a: f64 sqrt / divide / f32 rounding on the device must be correctly rounded: compare the
    variance tap against a numpy f64 recomputation from the oracle's integrals, bit for bit."""
    path = oracle.cascade_path("alt")
    casc = pkg.Cascade(path)
    frame = synth.texture(200, 260, synth.frame_seed(5))
    ctx = pkg.Context(casc)
    ctx.configure(200, 260, 1.1, 2, (0, 0), (0, 0), max_batch=1)
    ctx.detect_host(frame[None])
    for li, L in enumerate(ctx.levels()):
        got = ctx.read_tap(0, li, pkg.TAP_VNF)
        assert got.tobytes() == _vnf_want(oracle, casc, frame, L).tobytes(), "vnf bits differ at level %d" % li


def test_sync_call_matches_oracle_with_row_step(pkg, oracle, synth):
    """hc_detect_multiscale (the literal detectMultiScale replacement) on a non-contiguous view."""
    path = oracle.cascade_path("alt")
    big, _ = synth.planted(300, 420, synth.frame_seed(3), nfaces=3, min_size=40, max_size=120)
    view = big[10:290, 7:407]                       # step_bytes (420) != cols (400)
    got = pkg.Cascade(path).detect_multiscale(view, **CALL)
    want, _ = oracle.OracleCascade(path).detect(np.ascontiguousarray(view), 1.1, 2, (30, 30))
    assert np.array_equal(got, want)
    assert len(want) > 0


@pytest.mark.parametrize("shape", [(19, 40), (20, 20), (21, 23), (33, 31), (64, 47), (97, 131)])
def test_edge_sizes(pkg, oracle, synth, shape):
    """Smaller than the window (no levels), exactly the window, ragged sizes; minSize unset."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frame = synth.texture(shape[0], shape[1], synth.frame_seed(shape[0] * 1000 + shape[1]))
    got = casc.detect_multiscale(frame, 1.1, 0, 0, (0, 0))      # minNeighbors 0: raw candidates come back
    want, raw = oc.detect(frame, 1.1, 0, (0, 0))
    assert np.array_equal(got, want)
    if shape[0] >= 20 and shape[1] >= 20:
        ctx = pkg.Context(casc)
        ctx.configure(shape[0], shape[1], 1.1, 0, (0, 0), (0, 0), 1)
        ctx.enable_stage_taps(True)
        ctx.detect_host(frame[None])
        _check_frame_all_taps(pkg, oracle, ctx, oc, frame, 0, 1.1, (0, 0))


def test_constant_and_masked_frames(pkg, oracle, synth):
    """Constant image: every window has nf == 0 -> nothing evaluable, no detections.  Half-masked
    frame (the reference zeroes non-skin pixels, detector.cc:22): flat regions next to texture."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    const = np.full((120, 160), 77, np.uint8)
    assert len(casc.detect_multiscale(const, 1.1, 0, 0, (0, 0))) == 0
    fr, _ = synth.planted(240, 320, synth.frame_seed(9), nfaces=3, min_size=40, max_size=110)
    fr = fr.copy()
    fr[:, :130] = 0
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), 1)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(fr[None])
    grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (30, 30))
    assert np.array_equal(got[0], grouped)


def test_noise_skip_chain(pkg, oracle, synth):
    """iid noise fails stage 0 almost everywhere -> long x-skip chains across tile boundaries."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    fr = synth.noise(150, 700, synth.frame_seed(11))
    ctx = pkg.Context(casc)
    ctx.configure(150, 700, 1.1, 2, (0, 0), (0, 0), 1)
    ctx.enable_stage_taps(True)
    ctx.detect_host(fr[None])
    _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (0, 0))


@pytest.mark.parametrize("seg", [{"HC_SEG_MAX_BATCH": "0"}, {}, {"HC_SEG_TILES": "1"}, {"HC_SEG_TILES": "3"}])
def test_row_long_stage0_failure_chains(pkg, oracle, synth, monkeypatch, seg):
    """Horizontal stripes make EVERY window of a row evaluable and failing stage 0: the x-skip chain (A.7) then
    runs unbroken from the row start, which is the worst case for the segmented band walk of small batches (a
    segment must look back tile by tile to the row start to recover its skip state).  Left part stripes, right
    part noise and a constant block, so that chains also end and restart inside rows.  All taps, whole bands
    (HC_SEG_MAX_BATCH=0) and segments of 1, 2 (default) and 3 tiles."""
    for k, v in seg.items():
        monkeypatch.setenv(k, v)
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    H, W = 120, 900
    y, x = np.mgrid[0:H, 0:W]
    fr = np.where((y % 5) < 2, 228, 28).astype(np.uint8)                 # stripes: whole-row failure chains
    fr[:, 600:] = synth.noise(H, W - 600, synth.frame_seed(31))          # chains break up in noise
    fr[40:80, 300:420] = 128                                             # flat block: windows not evaluable
    ctx = pkg.Context(casc)
    ctx.configure(H, W, 1.1, 2, (0, 0), (0, 0), 1)
    ctx.enable_stage_taps(True)
    ctx.detect_host(fr[None])
    _, _, taps = oc.detect(fr, 1.1, 2, (0, 0), (0, 0), want_taps=True)
    runs = max(int(((t[r] == 0) | (t[r] == -2))[:128].all()) for t in taps[:3] for r in range(t.shape[0]))
    assert runs == 1, "stimulus lost its long failure chains"
    _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (0, 0))


def test_scale_factor_and_size_limits(pkg, oracle, synth):
    path = oracle.cascade_path("alt2")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    fr, _ = synth.planted(240, 320, synth.frame_seed(21), nfaces=4, min_size=40, max_size=110)
    for sf, mn, mins, maxs in [(1.2, 3, (0, 0), (0, 0)), (1.05, 2, (40, 40), (90, 90)), (1.3, 1, (24, 24), (0, 0)),
                               (1.1, 2, (200, 200), (200, 200))]:
        got = casc.detect_multiscale(fr, sf, mn, 0, mins, maxs)
        want, _ = oc.detect(fr, sf, mn, mins, maxs)
        assert np.array_equal(got, want), (sf, mn, mins, maxs)


def test_batch_determinism_and_device_input(pkg, oracle, synth):
    """Same frames through batch slots / repeated runs give identical bytes (stands in for a race
    detector on the device), and the device-resident entry point equals the host one."""
    path = oracle.cascade_path("alt")
    casc = pkg.Cascade(path)
    frames = np.stack([synth.planted(180, 240, synth.frame_seed(i), nfaces=2, min_size=40, max_size=80)[0]
                       for i in range(5)])
    ctx = pkg.Context(casc)
    ctx.configure(180, 240, 1.1, 2, (30, 30), (0, 0), max_batch=8, cand_capacity=1 << 14)
    a = ctx.detect_host(frames)
    b = ctx.detect_host(frames[::-1].copy())[::-1]
    c = ctx.detect_host(frames)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    single = [casc.detect_multiscale(f, **CALL) for f in frames]
    for x, y in zip(a, single):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("split", ["1", "0"])
def test_full_size_720p_alt(pkg, oracle, synth, monkeypatch, split):
    """BASELINE config #2 at full size: 1280x720, alt, sf 1.1, minNeighbors 2, minSize 30.
    Boxes + raw candidates + per-window stage counts against the oracle on one planted frame and
    one noise frame (oracle ~1 s per frame).  With the fused prep pinned this runs the column-strip kernel (default) and,
    with HC_VAR_SPLIT=0, round 3's one-workgroup-per-level kernel with its register and LDS rings (two levels of this ladder
    have more than 512 grid columns), which stays in the library as the A/B reference; the variance factors of four levels
    are compared as a tap of their own."""
    monkeypatch.setenv("HC_VAR_SPLIT", split)
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted(720, 1280, synth.frame_seed(0))[0], synth.noise(720, 1280, synth.frame_seed(1))])
    ctx = pkg.Context(casc)
    ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=2, cand_capacity=1 << 14)
    assert ctx.algorithmic_bytes() == 33813536                    # SURVEY 8d / BASELINE.md section 4
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frames)
    raws = ctx.raw_candidates(2)
    assert len(ctx.levels()) == 33
    for f in range(2):
        grouped, raw, otaps = oc.detect(frames[f], 1.1, 2, (30, 30), (0, 0), want_taps=True)
        for li in range(33):
            assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (f, li)
        assert np.array_equal(raws[f], raw)
        assert np.array_equal(got[f], grouped)
    assert len(got[0]) >= 5
    info = ctx.last_launch()
    if info["prep_route"] == "fused":
        assert info["variance_route"] == ("ring_strips" if split == "1" else "ring"), info
    for li in (0, 3, 4, 20):
        L = ctx.levels()[li]
        assert ctx.read_tap(1, li, pkg.TAP_VNF).tobytes() == _vnf_want(oracle, casc, frames[1], L).tobytes(), (split, li)


def test_cpp_shim_matches_oracle(pkg, oracle, synth, tmp_path):
    """The C++ cv::CascadeClassifier stand-in (hip_cascade/cascade_classifier.hpp), driven exactly
    like reference src/tracker/detector.cc:25-31, prints the oracle's boxes."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "shim_demo")
    frame, _ = synth.planted(240, 320, synth.frame_seed(0), nfaces=3, min_size=40, max_size=110)
    f = tmp_path / "frame.u8"
    frame.tofile(f)
    for name in ("alt", "alt2"):
        r = subprocess.run([demo, oracle.cascade_path(name), str(f), "240", "320", "1.1"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.array([[int(v) for v in ln.split()] for ln in r.stdout.splitlines()], np.int32).reshape(-1, 4)
        want, _ = oracle.OracleCascade(oracle.cascade_path(name)).detect(frame, 1.1, 2, (30, 30))
        assert np.array_equal(got, want) and len(want) > 0


def test_device_resident_input(pkg, oracle, synth):
    """hc_detect_batch_enqueue on frames that already live in HBM (torch only provides the memory)."""
    import torch
    path = oracle.cascade_path("alt")
    casc = pkg.Cascade(path)
    frames = np.stack([synth.planted(180, 240, synth.frame_seed(i), nfaces=2, min_size=40, max_size=80)[0]
                       for i in range(3)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ctx = pkg.Context(casc)
    ctx.configure(180, 240, 1.1, 2, (30, 30), (0, 0), max_batch=4)
    ctx.enqueue_device(d.data_ptr(), 3)
    got = ctx.wait(3)
    host = ctx.detect_host(frames)
    oc = oracle.OracleCascade(path)
    for f in range(3):
        want, _ = oc.detect(frames[f], 1.1, 2, (30, 30))
        assert np.array_equal(got[f], want) and np.array_equal(host[f], want)


def test_candidate_overflow_grows_and_reruns(pkg, oracle, synth):
    """A raw-candidate workspace that is too small is not an error the caller sees (OpenCV would simply return more
    rectangles): the wait grows the buffers to the exact need and runs the batch again.  Batched API with a capacity
    of 4 candidates, then the synchronous drop-in call on a frame whose minNeighbors-0 output exceeds its default."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted(240, 320, synth.frame_seed(i), nfaces=3, min_size=40, max_size=110)[0] for i in range(3)])
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=3, cand_capacity=4)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frames)
    raws = ctx.raw_candidates(3)
    again = ctx.detect_host(frames)                  # the grown workspace is kept: no second re-run needed
    for f in range(3):
        want, raw = oc.detect(frames[f], 1.1, 2, (30, 30))
        assert len(raw) > 4
        assert np.array_equal(got[f], want) and np.array_equal(raws[f], raw) and np.array_equal(again[f], want)
    _check_frame_all_taps(pkg, oracle, ctx, oc, frames[1], 1, 1.1, (30, 30))
    # Caller-owned device frames (include/hip_cascade.h, hc_detect_batch_enqueue): stream-ordered recycling of the
    # buffer is legal, so by default the library does NOT read it again -- the wait grows the workspace and reports
    # HC_ERR_CAND_OVERFLOW; the caller enqueues the batch again and gets every rectangle.
    import torch
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    want0 = [oc.detect(frames[f], 1.1, 0, (30, 30))[0] for f in range(3)]
    st = torch.cuda.Stream()
    c2 = pkg.Context(casc, stream=st.cuda_stream)
    c2.configure(240, 320, 1.1, 0, (30, 30), (0, 0), max_batch=3, cand_capacity=8)
    with torch.cuda.stream(st):
        c2.enqueue_device(d.data_ptr(), 3)
        d.zero_()                                    # the decoder's next frames: ordered behind the batch's kernels on the same stream
        with pytest.raises(pkg.HipCascadeError) as ei:
            c2.wait(3)
        assert ei.value.code == -8 and "enqueue the batch again" in str(ei.value)
        d.copy_(torch.from_numpy(frames).cuda(), non_blocking=False)
        c2.enqueue_device(d.data_ptr(), 3)           # the workspace has grown: this pass fits
        got0 = c2.wait(3)
    for f in range(3):
        assert np.array_equal(got0[f], want0[f])
    # ... and with the caller's promise to keep the frames unmodified until the wait returns, the library re-runs by itself
    c3 = pkg.Context(casc)
    c3.configure(240, 320, 1.1, 0, (30, 30), (0, 0), max_batch=3, cand_capacity=8)
    c3.set_rerun_on_overflow(True)
    c3.enqueue_device(d.data_ptr(), 3)
    got1 = c3.wait(3)
    for f in range(3):
        assert np.array_equal(got1[f], want0[f])


def test_color_input_is_converted_like_upstream(pkg, oracle, synth):
    """SURVEY 8b "Semantics": a multi-channel image handed to detectMultiScale is converted with
    cvtColor(BGR2GRAY) first.  hc_convert_bgr2gray bytes against the oracle (3 and 4 channels, ragged width, row
    step), then hc_detect_multiscale_image on BGR / BGRA frames against the oracle run on the converted gray."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    bgr, _ = synth.planted_bgr(241, 323, synth.frame_seed(7), nfaces=3, min_size=40, max_size=110)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, (57, 91, 3), dtype=np.uint8)
    for img in (bgr, noise):
        want = oracle.bgr2gray(img)
        assert np.array_equal(pkg.bgr2gray(img), want)
        bgra = np.concatenate([img, rng.integers(0, 256, img.shape[:2] + (1,), dtype=np.uint8)], axis=2)
        assert np.array_equal(pkg.bgr2gray(bgra), want)
        wide = np.zeros((img.shape[0], img.shape[1] + 5, 3), np.uint8)
        wide[:, :img.shape[1]] = img
        assert np.array_equal(pkg.bgr2gray(wide[:, :img.shape[1]]), want)          # row step > row bytes
    gray = oracle.bgr2gray(bgr)
    want, _ = oc.detect(gray, 1.1, 2, (30, 30))
    assert len(want) > 0
    assert np.array_equal(casc.detect_multiscale(bgr, **CALL), want)
    bgra = np.concatenate([bgr, np.full(bgr.shape[:2] + (1,), 255, np.uint8)], axis=2)
    assert np.array_equal(casc.detect_multiscale(bgra, **CALL), want)
    assert np.array_equal(casc.detect_multiscale(gray, **CALL), want)              # and the plain gray call after it


@pytest.mark.parametrize("src,scan", [((720, 1280), (405, 720)), ((480, 854), (360, 640)), ((500, 700), (360, 640))])
def test_scan_resize_any_ratio(pkg, oracle, synth, src, scan):
    """reference src/tracker/face_tracker.cc:112-114 resizes to whatever `scan.frame_size` says: INTER_AREA with
    non-integer ratios (equal and unequal in x and y) on device against the oracle, then detectFaces on the result."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted_bgr(src[0], src[1], synth.frame_seed(60 + i), nfaces=5, min_size=80, max_size=220)[0]
                       for i in range(2)])
    ctx = pkg.Context(casc)
    ctx.configure(scan[0], scan[1], 1.1, 2, (30, 30), (0, 0), max_batch=2, cand_capacity=1 << 14)
    ctx.enable_faces()
    ctx.enable_scan_resize(src[0], src[1])
    got = ctx.faces_host(frames)
    for f in range(2):
        small = oracle.resize_area(frames[f], scan[0], scan[1])
        assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_BGR), small)
        want, mask, gray = oc.detect_faces(small, want_taps=True)
        assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_MASK), mask)
        assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_GRAY), gray)
        assert np.array_equal(got[f], want)
    with pytest.raises(pkg.HipCascadeError) as e:
        ctx.enable_scan_resize(scan[0] - 1, src[1])          # enlarging in y: not INTER_AREA's shrinking path
    assert e.value.code == -4


def _truncated_cascade(oracle, tmp_path, name, nstages, poison_leaf, poison_weight=False):
    """Write an HCC1 file with the first `nstages` stages of a reference cascade; optionally make
    one deep-stage leaf tiny so that the stage's f64 sum is no longer provably exact."""
    import struct
    import np_oracle_path  # noqa: F401
    import np_oracle as npo
    c = npo.load_hcc(oracle.cascade_path(name))
    ntr = int(c.stage_ntrees[:nstages].sum())
    nnd = int(c.tree_nnodes[:ntr].sum())
    nlf = nnd + ntr
    leaves = c.leaves[:nlf].copy()
    if poison_leaf:
        leaves[nlf - 3] = np.float32(1e-30)
    weights = c.weights.copy()
    if poison_weight:
        weights[5, 1] = np.float32(2.5)          # a non-integer rect weight: the exact-integer feature build must not be used
        weights[nnd // 2, 0] = np.float32(-2.0)  # and a first weight other than -1
    out = bytearray(b"HCC1") + struct.pack("<7i", c.win_w, c.win_h, nstages, ntr, nnd, nlf, 0)
    for i in range(nstages):
        out += struct.pack("<d3i", float(c.stage_thr_raw[i]), int(c.stage_ntrees[i]), -1, -1)
    for i in range(ntr):
        out += struct.pack("<i", int(c.tree_nnodes[i]))
    for i in range(nnd):
        out += struct.pack("<12b", *[int(v) for v in c.rects[i].ravel()])
        out += struct.pack("<3f", *[float(v) for v in weights[i]])
        out += struct.pack("<f3i", float(c.node_thr[i]), int(c.node_left[i]), int(c.node_right[i]), 0)
    out += leaves.astype("<f4").tobytes()
    p = tmp_path / ("%s_%d_%d_%d.hcc" % (name, nstages, int(poison_leaf), int(poison_weight)))
    p.write_bytes(bytes(out))
    return str(p)


@pytest.mark.parametrize("name", ["alt", "alt2"])
@pytest.mark.parametrize("poison", [False, True])
def test_sparse_mode_and_its_ordered_fallback(pkg, oracle, synth, tmp_path, name, poison):
    """Deep stages with few survivors run stump-parallel with an order-free f64 reduction, which is
    only legal when the stage's leaf sum is exact in f64.  A cascade whose last stage carries a
    1e-30 leaf breaks that proof: the kernel must take the ordered path there and still match."""
    path = _truncated_cascade(oracle, tmp_path, name, 8, poison)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    fr, _ = synth.planted(240, 320, synth.frame_seed(4), nfaces=3, min_size=40, max_size=110)
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), 1, 1 << 15)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(fr[None])
    grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (30, 30))
    assert np.array_equal(ctx.raw_candidates(1)[0], raw) and np.array_equal(got[0], grouped)
    assert len(raw) > 0


@pytest.mark.parametrize("name", ["alt", "alt2"])
def test_non_integer_rect_weights_take_the_f32_build(pkg, oracle, synth, tmp_path, name):
    """The production kernel forms feature values with 24-bit integer multiply-adds when the host has shown that every
    rect weight is a small integer with w0 == -1 (true for the four reference cascades).  A cascade with a weight of
    2.5 and a first weight of -2 must fall back to the f32 arithmetic of upstream and still match the oracle."""
    path = _truncated_cascade(oracle, tmp_path, name, 8, False, poison_weight=True)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    fr, _ = synth.planted(240, 320, synth.frame_seed(4), nfaces=3, min_size=40, max_size=110)
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), 1, 1 << 15)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(fr[None])
    grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (30, 30))
    assert np.array_equal(ctx.raw_candidates(1)[0], raw) and np.array_equal(got[0], grouped)


@pytest.mark.parametrize("name", ["alt2", "default"])
def test_full_size_1080p_two_cascades(pkg, oracle, synth, name):
    """BASELINE config #3: 1920x1080, alt2 and default (24x24 window) on the same frame."""
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frame = synth.planted(1080, 1920, synth.frame_seed(2))[0]
    ctx = pkg.Context(casc)
    ctx.configure(1080, 1920, 1.1, 2, (30, 30), (0, 0), max_batch=1, cand_capacity=1 << 15)
    assert len(ctx.levels()) == 37
    assert ctx.algorithmic_bytes() == (76051696 if name == "alt2" else 110293312)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frame[None])
    grouped, raw, otaps = oc.detect(frame, 1.1, 2, (30, 30), (0, 0), want_taps=True)
    for li in range(37):
        assert np.array_equal(ctx.read_tap(0, li, pkg.TAP_STAGES), otaps[li]), li
    assert np.array_equal(ctx.raw_candidates(1)[0], raw)
    assert np.array_equal(got[0], grouped) and len(grouped) >= 4


def test_full_size_2160p(pkg, oracle, synth):
    """BASELINE config #5 geometry: one 3840x2160 frame (45 levels, 2384-pixel-wide level 0)."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frame = synth.planted(2160, 3840, synth.frame_seed(3), nfaces=8, min_size=60, max_size=500)[0]
    ctx = pkg.Context(casc)
    ctx.configure(2160, 3840, 1.1, 2, (30, 30), (0, 0), max_batch=1, cand_capacity=1 << 16)
    assert len(ctx.levels()) == 45 and ctx.algorithmic_bytes() == 303677664
    got = ctx.detect_host(frame[None])
    grouped, raw = oc.detect(frame, 1.1, 2, (30, 30), (0, 0))
    assert np.array_equal(ctx.raw_candidates(1)[0], raw)
    assert np.array_equal(got[0], grouped) and len(grouped) >= 4
    # integrals of the widest level and of a mid-size one against the oracle (fused routes: four resp. two column strips per row, the
    # carries handed across); the squares' integral only exists in global memory when a tap asks for it, so the frame runs
    # again with the taps on, which also checks the stage counts of those two levels
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frame[None])
    assert np.array_equal(got[0], grouped)
    _, _, otaps = oc.detect(frame, 1.1, 2, (30, 30), (0, 0), want_taps=True)
    for li in (0, 4):
        L = ctx.levels()[li]
        img = oracle.resize_linear_exact(frame, L.h, L.w)
        s, q = oracle.integral(img)
        assert np.array_equal(ctx.read_tap(0, li, pkg.TAP_PYRAMID), img), li
        assert np.array_equal(ctx.read_tap(0, li, pkg.TAP_SUM), s) and np.array_equal(ctx.read_tap(0, li, pkg.TAP_SQSUM), q), li
        assert np.array_equal(ctx.read_tap(0, li, pkg.TAP_STAGES), otaps[li]), li
        assert ctx.read_tap(0, li, pkg.TAP_VNF).tobytes() == _vnf_want(oracle, casc, frame, L).tobytes(), li


def test_unfused_prep_kernels_agree(pkg, oracle, synth, monkeypatch):
    """HC_FUSED_PREP=0 keeps the three-kernel pyramid/integral path alive (debug / A-B): same taps."""
    monkeypatch.setenv("HC_FUSED_PREP", "0")
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    fr, _ = synth.planted(240, 320, synth.frame_seed(1), nfaces=3, min_size=40, max_size=110)
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), 1, 1 << 14)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(fr[None])
    grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (30, 30))
    assert np.array_equal(got[0], grouped)


def test_prep_policy_by_batch_size(pkg, oracle, synth, monkeypatch):
    """Without HC_FUSED_PREP the context picks the prep kernels by batch size (parallel kernels up to 16
    720p-frame equivalents, fused k_pyramid_integral above; lowered here so that both sides are reached):
    both give the oracle's boxes and identical integral words."""
    monkeypatch.delenv("HC_FUSED_PREP", raising=False)
    monkeypatch.setenv("HC_FUSED_MIN_BATCH", "0.3")   # 12 frames of 180x240 = 0.56 equivalents: fused; 4 frames: not
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted(180, 240, synth.frame_seed(20 + i), nfaces=2, min_size=40, max_size=80)[0]
                       for i in range(12)])
    want = [oc.detect(f, 1.1, 2, (30, 30))[0] for f in frames]
    ctx = pkg.Context(casc)
    ctx.configure(180, 240, 1.1, 2, (30, 30), (0, 0), max_batch=12, cand_capacity=1 << 14)
    big = ctx.detect_host(frames)                      # 12 frames: fused
    sum_big = ctx.read_tap(3, 0, pkg.TAP_SUM)
    small = ctx.detect_host(frames[:4])                # 4 frames: parallel kernels
    sum_small = ctx.read_tap(3, 0, pkg.TAP_SUM)
    assert np.array_equal(sum_big, sum_small)
    for i in range(12):
        assert np.array_equal(oracle.sorted_boxes(big[i]), oracle.sorted_boxes(want[i]))
    for i in range(4):
        assert np.array_equal(small[i], big[i])
    one = casc.detect_multiscale(frames[5], **CALL)    # the synchronous drop-in call: one frame
    assert np.array_equal(one, big[5])


@pytest.mark.parametrize("fused", ["1", "0"])
def test_detect_faces_path(pkg, oracle, synth, monkeypatch, fused):
    """FaceDetector::detectFaces (reference src/tracker/detector.cc:10-43) on device --
    skin mask, equalised masked gray, the cascade call and the skin-ratio post-filter -- against the
    oracle's restatement: mask bytes, gray bytes and the surviving rectangles are bit-exact."""
    monkeypatch.setenv("HC_FUSED_PREP", fused)   # mask integral by the fused kernel / by the row + column scans
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted_bgr(240, 320, synth.frame_seed(i), nfaces=3, min_size=40, max_size=110)[0]
                       for i in range(3)])
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=4, cand_capacity=1 << 14)
    ctx.enable_faces()
    got = ctx.faces_host(frames)
    nkept = 0
    for f in range(3):
        want, mask, gray = oc.detect_faces(frames[f], want_taps=True)
        assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_MASK), mask)
        assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_GRAY), gray)
        assert np.array_equal(got[f], want)
        nkept += len(want)
        # the unfiltered call on the same gray finds at least as many boxes
        allb, _ = oc.detect(gray, 1.1, 2, (30, 30))
        assert len(allb) >= len(want)
    assert nkept > 0
    # synchronous single-frame form, custom bounds, non-contiguous rows
    big = np.zeros((240, 330, 3), np.uint8)
    big[:, :320] = frames[0]
    view = big[:, :320]
    lo, hi = (10, 135, 80), (250, 170, 125)
    assert np.array_equal(casc.detect_faces(view, lo, hi, 1.1), oc.detect_faces(frames[0], lo, hi, 1.1))
    const = np.full((120, 160, 3), 90, np.uint8)
    assert len(casc.detect_faces(const)) == 0


def test_cpp_face_detector_mirror(pkg, oracle, synth, tmp_path):
    """hip_cascade::FaceDetector (mirror of ugproj::FaceDetector, reference src/tracker/detector.h:10-19)
    through shim_demo --faces: same rectangles as the oracle's detectFaces."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "shim_demo")
    bgr, _ = synth.planted_bgr(240, 320, synth.frame_seed(1), nfaces=3, min_size=40, max_size=110)
    f = tmp_path / "frame.bgr"
    bgr.tofile(f)
    r = subprocess.run([demo, "--faces", oracle.cascade_path("alt"), str(f), "240", "320"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(v) for v in ln.split()] for ln in r.stdout.splitlines()], np.int32).reshape(-1, 4)
    want = oracle.OracleCascade(oracle.cascade_path("alt")).detect_faces(bgr)
    assert np.array_equal(got, want) and len(want) > 0


def test_config1_plumbing_scan_resize_then_detect_faces(pkg, oracle, synth):
    """BASELINE config #1 (conf/apink.mrchu.720p.cfg as shipped: scan 640x360, cascade alt2): 1280x720 BGR
    frames -> INTER_AREA resize to the scan size (reference src/tracker/face_tracker.cc:112-114) ->
    FaceDetector::detectFaces (src/tracker/detector.cc:10-43).  Every stage against the oracle."""
    for name in ("alt2", "alt"):           # as shipped (cfg:10), and with the cascade BASELINE.json names
        path = oracle.cascade_path(name)
        casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
        frames = np.stack([synth.planted_bgr(720, 1280, synth.frame_seed(i), nfaces=6, min_size=90, max_size=260)[0]
                           for i in range(2)])
        ctx = pkg.Context(casc)
        ctx.configure(360, 640, 1.1, 2, (30, 30), (0, 0), max_batch=2, cand_capacity=1 << 14)
        assert len(ctx.levels()) == 26 and ctx.algorithmic_bytes() == 8464320
        ctx.enable_faces()
        ctx.enable_scan_resize(720, 1280)
        got = ctx.faces_host(frames)
        n = 0
        for f in range(2):
            small = oracle.resize_area(frames[f], 360, 640)
            assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_BGR), small)
            want, mask, gray = oc.detect_faces(small, want_taps=True)
            assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_MASK), mask)
            assert np.array_equal(ctx.read_face_tap(f, pkg.TAP_FACE_GRAY), gray)
            assert np.array_equal(got[f], want)
            n += len(want)
        assert n > 0
    ctx.enable_scan_resize(700, 1280)               # 700/360 is not an integer factor: the general path takes it


def test_randomised_calls_match_oracle(pkg, oracle, synth):
    """Property-style sweep: random geometry, scale factor, minNeighbors, min/max size, cascade and
    stimulus family; grouped boxes and raw candidates must equal the oracle's every time."""
    rng = np.random.default_rng(2026)
    cascades = {n: (pkg.Cascade(oracle.cascade_path(n)), oracle.OracleCascade(oracle.cascade_path(n)))
                for n in ("alt", "alt2", "default", "alt_tree")}
    nboxes = 0
    for it in range(40):
        name = ("alt", "alt2", "default", "alt_tree")[it % 4]
        casc, oc = cascades[name]
        rows, cols = int(rng.integers(24, 300)), int(rng.integers(24, 400))
        kind = it % 3
        seed = synth.frame_seed(1000 + it)
        if kind == 0 and rows >= 60 and cols >= 60:
            fr = synth.planted(rows, cols, seed, nfaces=2, min_size=30, max_size=max(31, min(rows, cols) - 1))[0]
        elif kind == 1:
            fr = synth.texture(rows, cols, seed)
        else:
            fr = synth.noise(rows, cols, seed)
        sf = float(rng.choice([1.05, 1.1, 1.15, 1.2, 1.3, 1.5, 2.0]))
        mn = int(rng.integers(0, 4))
        mins = (int(rng.integers(0, 50)),) * 2 if rng.integers(0, 2) else (0, 0)
        maxs = (int(rng.integers(60, 200)),) * 2 if rng.integers(0, 3) == 0 else (0, 0)
        ctx = pkg.Context(casc)
        ctx.configure(rows, cols, sf, mn, mins, maxs, max_batch=1, cand_capacity=1 << 16)
        got = ctx.detect_host(fr[None])[0]
        raw = ctx.raw_candidates(1)[0]
        want, wraw = oc.detect(fr, sf, mn, mins, maxs)
        assert np.array_equal(raw, wraw), (it, name, rows, cols, sf, mn, mins, maxs)
        assert np.array_equal(got, want), (it, name, rows, cols, sf, mn, mins, maxs)
        nboxes += len(want)
        ctx.close()
    assert nboxes > 0


def test_two_contexts_two_threads(pkg, oracle, synth):
    """A cascade is immutable and shareable; contexts are single-caller.  Two host threads, each
    with its own context on the same GPU, produce the same boxes as a serial run."""
    import threading
    path = oracle.cascade_path("alt")
    casc = pkg.Cascade(path)
    frames = np.stack([synth.planted(180, 240, synth.frame_seed(i), nfaces=2, min_size=40, max_size=80)[0]
                       for i in range(6)])
    ref = pkg.Context(casc)
    ref.configure(180, 240, 1.1, 2, (30, 30), (0, 0), max_batch=8)
    want = ref.detect_host(frames)
    results, errors = {}, []

    def work(tid):
        try:
            c = pkg.Context(casc)
            c.configure(180, 240, 1.1, 2, (30, 30), (0, 0), max_batch=8)
            for _ in range(5):
                results[tid] = c.detect_host(frames)
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for tid in range(2):
        for a, b in zip(results[tid], want):
            assert np.array_equal(a, b)


def test_context_lifecycle_soak(pkg, oracle, synth):
    """Create / configure / reconfigure / destroy many contexts: no leaks that exhaust HBM, no
    stale state (every configuration still matches the oracle)."""
    import torch
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    free0 = torch.cuda.mem_get_info()[0]
    rng = np.random.default_rng(5)
    ctx = pkg.Context(casc)
    for it in range(60):
        rows, cols = int(rng.integers(40, 400)), int(rng.integers(40, 500))
        if it % 10 == 0:
            ctx.close()
            ctx = pkg.Context(casc)
        ctx.configure(rows, cols, 1.1, 2, (0, 0), (0, 0), max_batch=int(rng.integers(1, 9)))
        if it % 3 == 0:
            ctx.enable_faces()
        fr = synth.noise(rows, cols, synth.frame_seed(it))
        if it % 6 == 0:
            want, _ = oc.detect(fr, 1.1, 2, (0, 0))
            assert np.array_equal(ctx.detect_host(fr[None])[0], want)
        else:
            ctx.detect_host(fr[None])
    ctx.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 256 << 20, "workspace memory is not returned: %d MiB lost" % ((free0 - free1) >> 20)


@pytest.mark.parametrize("pool", ["pageable", "pinned", "registered"])
def test_cpp_throughput_demo(pkg, oracle, synth, tmp_path, pool):
    """The batched C ABI from plain C++ (hip_cascade/throughput_demo.cpp): pipelined host-frame batches from
    pageable memory, from an hc_host_alloc pool and from a buffer pinned with hc_host_register;
    deterministic box count equal to the oracle's."""
    import os
    import subprocess
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "throughput_demo")
    frames = np.stack([synth.planted(240, 320, synth.frame_seed(i), nfaces=3, min_size=40, max_size=110)[0]
                       for i in range(10)])
    f = tmp_path / "frames.u8"
    frames.tofile(f)
    r = subprocess.run([demo, oracle.cascade_path("alt"), str(f), "10", "240", "320", "4", "3", pool],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    oc = oracle.OracleCascade(oracle.cascade_path("alt"))
    want = sum(len(oc.detect(fr, 1.1, 2, (30, 30))[0]) for fr in frames)
    assert ("%d boxes per pass" % want) in r.stdout, r.stdout


@pytest.mark.parametrize("knobs", [
    {"HC_BAND_ROWS1": "8", "HC_BAND_ROWS2": "16"},
    {"HC_BAND_ROWS1": "32"},
    {"HC_ROW_STAGES": "3", "HC_SPARSE_MAX": "0"},       # never sparse, three row-form stages
    {"HC_ROW_STAGES": "1"},                             # only stage 0 in row form (the round-1 default)
    {"HC_SPARSE_MAX": "512"},                           # always sparse once a tile has survivors
    {"HC_HOST_THREADS": "1", "HC_FUSED_PREP": "0"},
    {"HC_SEG_MAX_BATCH": "0", "HC_SIDE_MAX_BATCH": "0"},  # small batches without band segments / side stream
    {"HC_SEG_TILES": "1"},
    {"HC_FUSE_VARIANCE": "0", "HC_FUSED_PREP": "1"},    # fused prep kernel, variance factors by the separate k_variance
    {"HC_INT_WEIGHTS": "0"},                            # f32 feature arithmetic although the cascade qualifies for the integer build
    {"HC_BANK_ORDER": "0"},                             # sparse finish reads the stumps in the dense path's order
    {"HC_BAND_WAVES1": "8", "HC_BAND_WAVES2": "4"},     # swapped workgroup shapes
    {"HC_BAND_WAVES1": "8", "HC_BAND_ROWS1": "32", "HC_BAND_WAVES2": "4", "HC_BAND_ROWS2": "8"},
    {"HC_VAR_SPLIT": "0", "HC_FUSED_PREP": "1"},        # the one-workgroup-per-level prep kernel (round 3's) with its rings
    {"HC_VAR_SPLIT": "0", "HC_VAR_RING": "0", "HC_FUSED_PREP": "1"},   # ... and with its read-back route
])
def test_tuning_knobs_do_not_change_results(pkg, oracle, synth, monkeypatch, knobs):
    """Every tuning knob (read when a context is created) must leave all taps bit-exact."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for name in ("alt", "alt2"):
        path = oracle.cascade_path(name)
        casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
        fr, _ = synth.planted(240, 320, synth.frame_seed(12), nfaces=3, min_size=40, max_size=110)
        ctx = pkg.Context(casc)
        ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), 1, 1 << 14)
        ctx.enable_stage_taps(True)
        got = ctx.detect_host(fr[None])
        grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, fr, 0, 1.1, (30, 30))
        assert np.array_equal(ctx.raw_candidates(1)[0], raw) and np.array_equal(got[0], grouped)


@pytest.mark.parametrize("chunks", [("8", "8"), ("16", "0"), ("0", "16")])
def test_frame_chunks_cover_every_frame_and_band(pkg, oracle, synth, monkeypatch, chunks):
    """Large batches are walked in chunks of frames (k_cascade: all bands of 128 frames before the next 128; k_pyramid_integral:
    all levels of 32 frames) -- a remap of the workgroup index.  Small chunks (environment knobs) over a batch of 20 frames, which
    is two full chunks of 8 and a short one: stage taps of every frame and level, raw candidates and boxes."""
    monkeypatch.setenv("HC_FRAME_CHUNK", chunks[0])
    monkeypatch.setenv("HC_PI_FRAME_CHUNK", chunks[1])
    monkeypatch.setenv("HC_FUSED_PREP", "1")
    monkeypatch.setenv("HC_SEG_MAX_BATCH", "0")               # whole bands: the remap lives in the band kernel
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    kinds = ["planted", "texture", "noise"]
    frames = np.stack([synth.make_frames(kinds[i % 3], 1, 240, 320, first=1200 + i)[0] for i in range(20)])
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=20, cand_capacity=1 << 13)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frames)
    raws = ctx.raw_candidates(20)
    # The remaps must have RUN with these values: every knob is parsed into the context when it is created, and the
    # launch report carries what the launchers were actually given (round 3 cached HC_PI_FRAME_CHUNK in a function static,
    # so this test silently ran k_pyramid_integral with the value an earlier test of the process had set).
    info = ctx.last_launch()
    ccas, cpi = int(chunks[0]), int(chunks[1])
    assert info["prep_route"] == "fused" and info["nframes"] == 20, info
    assert (info["casc_frame_chunk"], info["casc_remapped"], info["casc_tail_frames"]) == (ccas, int(ccas > 0), 20 % ccas if ccas else 0), info
    assert (info["pi_frame_chunk"], info["pi_remapped"], info["pi_tail_frames"]) == (cpi, int(cpi > 0), 20 % cpi if cpi else 0), info
    assert info["segments"] == 0 and info["variant"] == ("p16", "spec"), info
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as ex:             # the oracle call releases the GIL
        want = list(ex.map(lambda fr: oc.detect(fr, 1.1, 2, (30, 30), (0, 0), want_taps=True), frames))
    for f in range(20):
        grouped, raw, otaps = want[f]
        for li in range(len(ctx.levels())):
            assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (chunks, f, li)
        assert np.array_equal(raws[f], raw) and np.array_equal(got[f], grouped), (chunks, f)
    ctx.enable_stage_taps(False)                              # the instantiation without taps, same remap
    got2 = ctx.detect_host(frames)
    for f in range(20):
        assert np.array_equal(got2[f], want[f][0]), (chunks, f)


@pytest.mark.parametrize("route,env", [("ring_strips", {}), ("kernel", {"HC_VAR_SPLIT": "0"}),
                                       ("readback", {"HC_VAR_SPLIT": "0", "HC_FUSE_VARIANCE": "1"})])
def test_wide_levels_variance_routes(pkg, oracle, synth, monkeypatch, route, env):
    """Levels wider than 1024 columns (1920x1080: 1192) have no ring route in the one-workgroup-per-level prep kernel.  By
    policy such a geometry runs the column-strip form (k_pyramid_strips: every grid column on a register ring, the carry
    handed from strip to strip); HC_VAR_SPLIT=0 restores round 3's choice -- k_variance behind the fused kernel -- and
    HC_FUSE_VARIANCE=1 with it the kernel's read-back route.  All three must give the same integrals, factors, taps and boxes."""
    monkeypatch.setenv("HC_FUSED_PREP", "1")
    for k in ("HC_VAR_SPLIT", "HC_FUSE_VARIANCE"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.planted(1080, 1920, synth.frame_seed(900 + i), nfaces=5, min_size=50, max_size=300)[0] for i in range(2)])
    ctx = pkg.Context(casc)
    ctx.configure(1080, 1920, 1.1, 2, (30, 30), (0, 0), 2, 1 << 15)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frames)
    info = ctx.last_launch()
    assert info["prep_route"] == "fused" and info["variance_route"] == route, info
    for f in range(2):
        grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, frames[f], f, 1.1, (30, 30))
        assert np.array_equal(ctx.raw_candidates(2)[f], raw) and np.array_equal(got[f], grouped)
    # the variance factors themselves (a tap of their own): every scanned level of frame 1
    for li, L in enumerate(ctx.levels()):
        if L.nx and L.ny:
            assert ctx.read_tap(1, li, pkg.TAP_VNF).tobytes() == _vnf_want(oracle, casc, frames[1], L).tobytes(), (route, li)


@pytest.mark.parametrize("name,shape,comp", [("alt", (240, 320), 96), ("default", (240, 320), 64), ("alt2", (203, 331), 128),
                                             ("alt_tree", (360, 640), 160), ("alt", (720, 1280), 224)])
def test_column_strips_on_small_frames(pkg, oracle, synth, monkeypatch, name, shape, comp):
    """The column-strip form of the fused prep kernel with NARROW strips (HC_STRIP_COMP_MAX) on small frames, whose levels
    a strip of the default width would take whole: every level then runs through two to six strips, so the carry
    hand-off, the halo columns, ragged last chunks (heights that are no multiple of eight), strips without grid columns and
    both inner-rect heights (18: alt / alt2 / alt_tree, 22: default) are all exercised at sizes the oracle finishes in
    seconds.  Every tap of every level -- pyramid bytes, both integrals (sqsum: the tap asks for it), stage counts -- then
    raw candidates and boxes, for a batch of five frames of mixed families."""
    monkeypatch.setenv("HC_FUSED_PREP", "1")
    monkeypatch.setenv("HC_STRIP_COMP_MAX", str(comp))
    monkeypatch.setenv("HC_PI_FRAME_CHUNK", "0" if comp == 96 else "8")
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    kinds = ["planted", "texture", "noise"]
    B = 5 if shape[0] < 700 else 10
    frames = np.stack([synth.make_frames(kinds[i % 3], 1, shape[0], shape[1], first=4100 + i)[0] for i in range(B)])
    ctx = pkg.Context(casc)
    ctx.configure(shape[0], shape[1], 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 14)
    ctx.enable_stage_taps(True)
    got = ctx.detect_host(frames)
    info = ctx.last_launch()
    assert info["prep_route"] == "fused" and info["variance_route"] == "ring_strips", info
    raws = ctx.raw_candidates(B)
    for f in range(B):
        if f in (0, B - 1):
            grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, frames[f], f, 1.1, (30, 30))
            for li, L in enumerate(ctx.levels()):
                if L.nx and L.ny:
                    assert ctx.read_tap(f, li, pkg.TAP_VNF).tobytes() == _vnf_want(oracle, casc, frames[f], L).tobytes(), (name, f, li)
        else:
            grouped, raw = oc.detect(frames[f], 1.1, 2, (30, 30), (0, 0))
        assert np.array_equal(raws[f], raw) and np.array_equal(got[f], grouped), (name, f)
    ctx.enable_stage_taps(False)           # the route the bench takes: no pyramid, no squares' integral in global memory
    got2 = ctx.detect_host(frames)
    assert ctx.last_launch()["sq_stored"] == 0
    for f in range(B):
        assert np.array_equal(got2[f], got[f]), (name, f)


def test_strip_handoff_under_load(pkg, oracle, synth, monkeypatch):
    """The strip-to-strip hand-off (plain sc1 stores of the carries, an sc1 row count, sc1 polls and loads on the other side; no
    fences) with the chip oversubscribed and unevenly loaded: 24 noise frames of 1920x1080 in one batch with 256-column strips --
    up to five strips per level, about 5,000 workgroups for 512 resident slots, neighbours on different XCDs, consumers that have
    polled the same line many times.  EVERY word of both integrals of three multi-strip levels of EVERY frame against the oracle
    (a stale or early carry shifts a whole strip of a row), then boxes of all frames."""
    monkeypatch.setenv("HC_FUSED_PREP", "1")
    monkeypatch.setenv("HC_STRIP_COMP_MAX", "256")
    monkeypatch.setenv("HC_PI_FRAME_CHUNK", "8")
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    B = 24
    frames = np.stack([synth.make_frames("noise" if i % 2 else "texture", 1, 1080, 1920, first=5200 + i)[0] for i in range(B)])
    ctx = pkg.Context(casc)
    ctx.configure(1080, 1920, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 14)
    ctx.enable_stage_taps(True)                        # the squares' integral and the pyramid exist in global memory only for a tap
    got = ctx.detect_host(frames)
    info = ctx.last_launch()
    assert info["variance_route"] == "ring_strips" and info["pi_remapped"] == 1 and info["sq_stored"] == 1, info
    levels = ctx.levels()
    picks = [0, 3, 6]
    assert all(levels[li].w > 600 for li in picks)      # three to five 256-column strips each
    from concurrent.futures import ThreadPoolExecutor

    def want(args):
        f, li = args
        L = levels[li]
        return oracle.integral(oracle.resize_linear_exact(frames[f], L.h, L.w))
    jobs = [(f, li) for f in range(B) for li in picks]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        wants = list(ex.map(want, jobs))
    for (f, li), (ws, wq) in zip(jobs, wants):
        assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_SUM), ws), (f, li)
        assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_SQSUM), wq), (f, li)
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        boxes = list(ex.map(lambda fr: oc.detect(fr, 1.1, 2, (30, 30), (0, 0))[0], frames))
    for f in range(B):
        assert np.array_equal(got[f], boxes[f]), f


@pytest.mark.parametrize("name", ["alt", "default", "alt_tree", "alt2"])
@pytest.mark.parametrize("spec", ["1", "0"])
def test_compiled_early_stages_and_generic_kernel_agree(pkg, oracle, synth, monkeypatch, name, spec):
    """The stock cascades (three of stumps, alt2 of two-node trees) run their first stages as generated straight-line code (csrc/spec/); HC_SPEC=0 sends
    the same cascade through the generic kernel.  Both must reproduce every tap of the oracle, whole batches and
    band segments (one frame) alike, and the context must say which of the two actually ran."""
    monkeypatch.setenv("HC_SPEC", spec)
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    want_n = casc.compiled_stages()[1]
    assert want_n > 0
    frames = np.stack([synth.planted(360, 480, synth.frame_seed(40 + i), nfaces=4, min_size=40, max_size=160)[0] for i in range(3)])
    for batch in (3, 1):
        ctx = pkg.Context(casc)
        ctx.configure(360, 480, 1.1, 2, (30, 30), (0, 0), batch, 1 << 15)
        ctx.enable_stage_taps(True)
        got = ctx.detect_host(frames[:batch])
        assert ctx.compiled_stages() == (want_n if spec == "1" else 0)
        for i in range(batch):
            grouped, raw = _check_frame_all_taps(pkg, oracle, ctx, oc, frames[i], i, 1.1, (30, 30))
            assert np.array_equal(ctx.raw_candidates(batch)[i], raw) and np.array_equal(got[i], grouped)
        # and without the taps (the production instantiation of the kernel)
        ctx.enable_stage_taps(False)
        again = ctx.detect_host(frames[:batch])
        assert ctx.compiled_stages() == (want_n if spec == "1" else 0)
        for i in range(batch):
            assert np.array_equal(again[i], got[i])


def test_sync_contexts_are_cached_per_geometry(pkg, oracle, synth):
    """the synchronous drop-in call keeps a context per (geometry, call parameters), a few of them.  A
    tracker that alternates between two scan sizes allocates nothing after the first call of each size, and the
    alternating calls are as fast as calls that repeat one size (re-configuring used to cost ~20 hipMalloc per call)."""
    import time
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    a = synth.planted(360, 640, synth.frame_seed(11), nfaces=4, min_size=40, max_size=120)[0]
    b = synth.planted(720, 1280, synth.frame_seed(12))[0]
    want = {id(a): oc.detect(a, 1.1, 2, (30, 30))[0], id(b): oc.detect(b, 1.1, 2, (30, 30))[0]}
    for fr in (a, b):
        assert np.array_equal(casc.detect_multiscale(fr, 1.1, 2, 0, (30, 30)), want[id(fr)])
    n0 = pkg.alloc_count()
    assert n0 > 0
    alt = {id(a): [], id(b): []}
    for i in range(50):
        fr = a if i % 2 == 0 else b
        t0 = time.perf_counter()
        got = casc.detect_multiscale(fr, 1.1, 2, 0, (30, 30))
        alt[id(fr)].append(time.perf_counter() - t0)
        assert np.array_equal(got, want[id(fr)]), i
    assert pkg.alloc_count() == n0, "alternating between two cached geometries must not allocate"
    for fr in (a, b):
        steady = []
        for _ in range(25):
            t0 = time.perf_counter()
            casc.detect_multiscale(fr, 1.1, 2, 0, (30, 30))
            steady.append(time.perf_counter() - t0)
        m_alt, m_steady = float(np.median(alt[id(fr)])), float(np.median(steady))
        # the structural fact is the allocation count above; the timing is a sanity bound with room for a shared box (a
        # re-configuration costs tens of allocations and several milliseconds, i.e. far more than 2 x)
        assert m_alt <= 2.0 * m_steady + 300e-6, "alternating %.3f ms vs steady %.3f ms" % (1e3 * m_alt, 1e3 * m_steady)
    # other call parameters are another context; more combinations than the cache holds still give the right answers
    for k, (rows, cols) in enumerate([(240, 320), (300, 400), (360, 480), (200, 260), (250, 330), (240, 320)]):
        fr = synth.planted(rows, cols, synth.frame_seed(20 + k), nfaces=3, min_size=40, max_size=110)[0]
        assert np.array_equal(casc.detect_multiscale(fr, 1.2, 3, 0, (24, 24)), oc.detect(fr, 1.2, 3, (24, 24))[0]), (rows, cols)
    assert np.array_equal(casc.detect_multiscale(a, 1.1, 2, 0, (30, 30)), want[id(a)])


def test_capacity_error_keeps_the_rectangles(pkg, oracle, synth, tmp_path):
    """A too-small output buffer of a synchronous call: HC_ERR_CAPACITY, *out_count = the number found, and
    hc_detect_multiscale_reread hands the kept rectangles over (no second detection).  Then the C++ shim on a frame
    whose minNeighbors-0 output exceeds its first buffer of 1024."""
    import ctypes as C
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    fr = synth.planted(240, 320, synth.frame_seed(31), nfaces=3, min_size=40, max_size=110)[0]
    want, _ = oc.detect(fr, 1.1, 0, (30, 30))
    assert len(want) > 8
    small = np.zeros((2, 4), np.int32)
    n = C.c_int()
    rc = pkg.lib.hc_detect_multiscale(casc._h, fr.ctypes.data, 240, 320, fr.strides[0], 1.1, 0, 0, 30, 30, 0, 0, small.ctypes.data, 2, C.byref(n))
    assert rc == -7 and n.value == len(want)
    out = np.zeros((n.value, 4), np.int32)
    m = C.c_int()
    assert pkg.lib.hc_detect_multiscale_reread(casc._h, out.ctypes.data, n.value, C.byref(m)) == 0 and m.value == n.value
    assert np.array_equal(out, want)
    assert pkg.lib.hc_detect_multiscale_reread(casc._h, out.ctypes.data, n.value, C.byref(m)) == -9      # handed over once
    # legacy call: same contract
    want24, _ = oc.detect_cv24(fr, 1.1, 0, (30, 30))
    rc = pkg.lib.hc_detect_multiscale_cv24(casc._h, fr.ctypes.data, 240, 320, fr.strides[0], 1.1, 0, 2, 30, 30, 0, 0, small.ctypes.data, 2, C.byref(n))
    assert rc == -7 and n.value == len(want24)
    out = np.zeros((n.value, 4), np.int32)
    assert pkg.lib.hc_detect_multiscale_reread(casc._h, out.ctypes.data, n.value, C.byref(m)) == 0 and np.array_equal(out, want24)
    # C++ shim
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "shim_demo")
    big = synth.planted(720, 1280, synth.frame_seed(32), nfaces=70, min_size=40, max_size=90)[0]
    wantb, _ = oc.detect(big, 1.1, 0, (30, 30))
    assert len(wantb) > 1024, "stimulus no longer exceeds the shim's first buffer (%d)" % len(wantb)
    f = tmp_path / "big.u8"
    big.tofile(f)
    r = subprocess.run([demo, path, str(f), "720", "1280", "1.1", "modern", "1", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(v) for v in ln.split()] for ln in r.stdout.splitlines()], np.int32).reshape(-1, 4)
    assert np.array_equal(got, wantb)


def test_retrained_cascade_runs_its_plugin_stages(pkg, oracle, synth):
    """The shipped configuration selects haarcascade_frontalface_alt2 (reference conf/apink.mrchu.720p.cfg:10); a
    retrained file of that shape has another hash and used to fall back to the generic tree path.  With its plug-in
    registered (tools/build_spec_plugin.sh) a new context runs compiled stages for it; stage taps, raw candidates and
    boxes equal the oracle's evaluation of the SAME modified file, before and after, batch and single frame."""
    d = os.path.join(os.path.dirname(pkg.LIB_PATH), "build", "test_plugin")
    so, hcc = os.path.join(d, "libhc_spec_alt2mod.so"), os.path.join(d, "alt2_mod.hcc")
    assert os.path.exists(so) and os.path.exists(hcc), "make -C ugproj-detector_amd test-plugin (part of build())"
    oc = oracle.OracleCascade(hcc)
    frames = np.stack([synth.planted(360, 640, synth.frame_seed(50 + i), nfaces=4, min_size=40, max_size=120)[0] for i in range(20)])
    want = [oc.detect(fr, 1.1, 2, (30, 30), (0, 0), want_taps=True) for fr in frames[:3]]
    wantb = [oc.detect(fr, 1.1, 2, (30, 30))[0] for fr in frames]

    def run(casc):
        ctx = pkg.Context(casc)
        ctx.configure(360, 640, 1.1, 2, (30, 30), (0, 0), max_batch=20, cand_capacity=1 << 13)
        ctx.enable_stage_taps(True)
        got = ctx.detect_host(frames)
        raws = ctx.raw_candidates(20)
        for f in range(3):
            grouped, raw, otaps = want[f]
            for li in range(len(ctx.levels())):
                assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (f, li)
            assert np.array_equal(raws[f], raw)
        for f in range(20):
            assert np.array_equal(got[f], wantb[f]), f
        ctx.enable_stage_taps(False)
        got = ctx.detect_host(frames)                       # the production (no-tap) instantiation
        for f in range(20):
            assert np.array_equal(got[f], wantb[f]), f
        n = ctx.compiled_stages()
        ctx.close()
        return n
    if os.environ.get("HC_SPEC") == "0":
        pytest.skip("compiled stages switched off")
    casc = pkg.Cascade(hcc)
    already = casc.compiled_stages()[1] > 0                 # an earlier test of this process may have registered it
    if not already:
        assert run(casc) == 0                               # generic tree path
        pkg.load_spec_plugin(so)
    assert run(pkg.Cascade(hcc)) == 8                       # compiled stages from the plug-in
    assert np.array_equal(pkg.Cascade(hcc).detect_multiscale(frames[0], 1.1, 2, 0, (30, 30)), wantb[0])


@pytest.mark.parametrize("name", ["alt", "default", "alt_tree"])
def test_p16_build_and_standard_build_agree(pkg, oracle, synth, monkeypatch, name):
    """The stump cascades qualify for the 16-bit-tile build of the scan-step-1 launch (two windows per lane in the row-form
    stages; every rect sum formed modulo 2^16 is of a rect of at most 257 pixels: larger rects 0 are cut in two -- alt --,
    complemented by rect 1's side strips or, in checkerboard stumps, replaced by their quadrants -- default's 24 x 24 window,
    alt_tree): 20 frames of 720p, all families, as whole bands -- stage taps of every level of three frames, raw candidates and
    boxes of all, with the build on (default) and off."""
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    kinds = ["planted", "texture", "noise"]
    frames = np.stack([synth.make_frames(kinds[i % 3], 1, 720, 1280, first=700 + i)[0] for i in range(20)])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        want = list(ex.map(lambda fr: oc.detect(fr, 1.1, 2, (30, 30), (0, 0), want_taps=True), frames))
    for on in ("1", "0"):
        monkeypatch.setenv("HC_P16", on)
        ctx = pkg.Context(casc)
        ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=20, cand_capacity=1 << 13)
        for taps in (True, False):
            ctx.enable_stage_taps(taps)
            got = ctx.detect_host(frames)
            raws = ctx.raw_candidates(20)
            # which build ran is reported by the library, not assumed: a policy change that silently disabled the 16-bit
            # tile would otherwise leave this test green twice over the same kernel
            info = ctx.last_launch()
            assert info["variant"] == (("p16" if on == "1" else "spec"), "spec") and info["segments"] == 0, (on, taps, info)
            assert info["compiled_stages"][0] > 0 and info["compiled_stages"][1] > 0, info
            for f in range(20):
                grouped, raw, otaps = want[f]
                if taps and f in (0, 7, 14):
                    for li in range(len(ctx.levels())):
                        assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (on, f, li)
                assert np.array_equal(raws[f], raw), (on, taps, f)
                assert np.array_equal(got[f], grouped), (on, taps, f)
        ctx.close()


@pytest.mark.parametrize("name", ["default", "alt2", "alt"])
def test_trimmed_bands(pkg, oracle, synth, monkeypatch, name):
    """A band may cover one grid row less than the kernel's 16 where that row of the tile is what keeps one more workgroup off
    a CU (api_model.cpp, end of upload_model; HC_BAND_TRIM=0 switches it off): default's scan-step-2 launch runs 15-row bands
    (40,552 bytes of LDS, four workgroups per CU, instead of 41,768 and three).  Whatever the library decides for a cascade, the
    decision must be the one its rule states -- the launch report carries band rows and LDS bytes -- and every stage tap, raw
    candidate and box must be the oracle's with the trim on and off (the last band of a level, the x-skip carry and the
    segment walk of a single frame all see the shorter bands)."""
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.make_frames(["planted", "texture", "noise"][i % 3], 1, 720, 1280, first=900 + i)[0] for i in range(6)])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        want = list(ex.map(lambda fr: oc.detect(fr, 1.1, 2, (30, 30), (0, 0), want_taps=True), frames))
    steps = {}
    for on in ("1", "0"):
        monkeypatch.setenv("HC_BAND_TRIM", on)
        for batch in (6, 1):                                 # whole bands; band segments (one frame)
            ctx = pkg.Context(casc)
            ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=batch, cand_capacity=1 << 13)
            ctx.enable_stage_taps(True)
            got = ctx.detect_host(frames[:batch])
            raws = ctx.raw_candidates(batch)
            info = ctx.last_launch()
            steps[(on, batch)] = info["band_step"]
            for k in range(2):
                assert info["band_step"][k] in ((15, 16) if on == "1" else (16,)), info
                if info["band_step"][k] == 15:               # the rule: the trim bought a workgroup per CU
                    waves = 4 if k == 0 else 8
                    per_cu = min(160 * 1024 // info["lds_bytes"][k], 32 // waves)
                    assert per_cu * info["lds_bytes"][k] <= 160 * 1024 and per_cu > 1, info
            for f in range(batch):
                grouped, raw, otaps = want[f]
                for li in range(len(ctx.levels())):
                    assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (on, batch, f, li)
                assert np.array_equal(raws[f], raw) and np.array_equal(got[f], grouped), (on, batch, f)
            ctx.close()
    if name == "default":
        assert steps[("1", 6)][1] == 15 and steps[("0", 6)] == (16, 16), steps
