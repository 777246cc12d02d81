"""GPU parity of the legacy mode ("T2"): OpenCV 2.4.x semantics of the reference's call at
src/tracker/detector.cc:25-31 -- the generation it was built against (src/tracker/associator.cc:3) and the branch its
CASCADE_SCALE_IMAGE flag selects -- through hc_detect_multiscale_cv24, against the CPU oracle's restatement
(oracle/haar_oracle.cpp orc_detect_cv24, itself cross-checked against numpy in tests/test_oracle.py).
Bit-exact: resize bytes, per-window stage taps, raw candidates in scan order, grouped boxes.  Unpinned against a real
2.4 build, like the >= 4 semantics."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cv24_resize_bytes(pkg, oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    for dst in [(97, 131), (60, 81), (45, 61), (33, 47), (20, 27), (7, 9)]:
        assert np.array_equal(pkg.resize_linear_cv24(img, *dst), oracle.resize_linear_cv24(img, *dst)), dst
    even = rng.integers(0, 256, (80, 120), dtype=np.uint8)
    assert np.array_equal(pkg.resize_linear_cv24(even, 40, 60), oracle.resize_linear_cv24(even, 40, 60))       # exact 1/2: area kernel
    wide = np.zeros((97, 140), np.uint8)
    wide[:, :131] = img
    assert np.array_equal(pkg.resize_linear_cv24(wide[:, :131], 45, 61), oracle.resize_linear_cv24(img, 45, 61))   # row step > cols


@pytest.mark.parametrize("name", ["alt", "alt2", "default", "alt_tree"])
def test_cv24_all_taps_small(pkg, oracle, synth, name):
    """stumps, depth-2 trees, the 24x24 window and alt_tree's branching stage tree on planted / texture / noise frames"""
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = [synth.planted(240, 320, synth.frame_seed(300), nfaces=3, min_size=40, max_size=110)[0],
              synth.texture(150, 211, synth.frame_seed(301)), synth.noise(90, 120, synth.frame_seed(302))]
    nraw = 0
    for fr in frames:
        for mn in (2, 0):
            want, raw, wtaps = oc.detect_cv24(fr, 1.1, mn, (30, 30), (0, 0), want_taps=True)
            got, taps = casc.detect_multiscale_cv24(fr, 1.1, mn, 2, (30, 30), (0, 0), want_taps=True)
            assert len(taps) == len(wtaps) and casc.plan_cv24(*fr.shape, 1.1, (30, 30)) == oc.plan_cv24(*fr.shape, 1.1, (30, 30))
            for li, (a, b) in enumerate(zip(taps, wtaps)):
                assert np.array_equal(a, b), "%s level %d: %d taps differ" % (name, li, int((a != b).sum()))
            assert np.array_equal(got, want), (name, mn)
            if mn == 0:
                assert np.array_equal(got, raw)
                nraw += len(raw)
    assert nraw > 0


def test_cv24_full_size_720p_and_limits(pkg, oracle, synth):
    """the reference's literal call (scale 1.1, minNeighbors 2, 0|CASCADE_SCALE_IMAGE, Size(30,30)) on a 1280x720 frame,
    alt and alt_tree; other scale factors / size limits; the branch without CASCADE_SCALE_IMAGE is not restated"""
    fr = synth.planted(720, 1280, synth.frame_seed(0))[0]
    for name in ("alt", "alt_tree"):
        path = oracle.cascade_path(name)
        casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
        want, raw = oc.detect_cv24(fr, 1.1, 2, (30, 30))
        assert np.array_equal(casc.detect_multiscale_cv24(fr, 1.1, 2, 2, (30, 30)), want) and len(want) >= 4
        assert np.array_equal(casc.detect_multiscale_cv24(fr, 1.1, 0, 2, (30, 30), capacity=1 << 18), raw)
    casc, oc = pkg.Cascade(oracle.cascade_path("alt2")), oracle.OracleCascade(oracle.cascade_path("alt2"))
    small = synth.planted(240, 320, synth.frame_seed(21), nfaces=4, min_size=40, max_size=110)[0]
    for sf, mn, mins, maxs in [(1.2, 3, (0, 0), (0, 0)), (1.05, 2, (40, 40), (90, 90)), (1.3, 1, (24, 24), (0, 0))]:
        assert np.array_equal(casc.detect_multiscale_cv24(small, sf, mn, 2, mins, maxs), oc.detect_cv24(small, sf, mn, mins, maxs)[0]), (sf, mn)
    with pytest.raises(pkg.HipCascadeError) as e:
        casc.detect_multiscale_cv24(small, 1.1, 2, 0, (30, 30))
    assert e.value.code == -4
    # the two generations really differ on the same frame (different ladders, resize, grids, thresholds)
    t4 = casc.detect_multiscale(small, 1.1, 2, 2, (30, 30))
    t2 = casc.detect_multiscale_cv24(small, 1.1, 2, 2, (30, 30))
    assert len(t4) > 0 and len(t2) > 0


def test_cv24_through_the_cpp_shim(pkg, oracle, synth, tmp_path):
    """hip_cascade::CascadeClassifier::setLegacySemantics(true) + the reference's call"""
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "shim_demo")
    frame, _ = synth.planted(240, 320, synth.frame_seed(0), nfaces=3, min_size=40, max_size=110)
    f = tmp_path / "frame.u8"
    frame.tofile(f)
    r = subprocess.run([demo, oracle.cascade_path("alt_tree"), str(f), "240", "320", "1.1", "legacy"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(v) for v in ln.split()] for ln in r.stdout.splitlines()], np.int32).reshape(-1, 4)
    want, _ = oracle.OracleCascade(oracle.cascade_path("alt_tree")).detect_cv24(frame, 1.1, 2, (30, 30))
    assert np.array_equal(got, want) and len(want) > 0


def test_cv24_colour_input_through_the_cpp_shim(pkg, oracle, synth, tmp_path):
    """Legacy semantics with a colour image: 2.4's detectMultiScale converts BGR -> gray and then evaluates the legacy
    path; the shim does the same (it used to fall through to the >= 4 evaluation without saying so)."""
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "shim_demo")
    bgr, _ = synth.planted_bgr(240, 320, synth.frame_seed(2), nfaces=3, min_size=40, max_size=110)
    f = tmp_path / "frame.bgr"
    bgr.tofile(f)
    oc = oracle.OracleCascade(oracle.cascade_path("alt"))
    gray = oracle.bgr2gray(bgr)
    want24, _ = oc.detect_cv24(gray, 1.1, 2, (30, 30))
    want4, _ = oc.detect(gray, 1.1, 2, (30, 30))
    for mode, want in (("legacy", want24), ("modern", want4)):
        r = subprocess.run([demo, oracle.cascade_path("alt"), str(f), "240", "320", "1.1", mode, "3"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.array([[int(v) for v in ln.split()] for ln in r.stdout.splitlines()], np.int32).reshape(-1, 4)
        assert np.array_equal(got, want) and len(want) > 0, mode
    assert not np.array_equal(want24, want4)


@pytest.mark.parametrize("route", ["separate", "fused", "fused_sweep"])
@pytest.mark.parametrize("name", ["alt", "alt2", "default", "alt_tree"])
def test_cv24_batched_context(pkg, oracle, synth, name, route, monkeypatch):
    """The 2.4 evaluation as a batched, device-resident path (hc_context_set_semantics): 6 frames of mixed families in one
    batch -- the LDS-tiled band kernel with 2.4's node arithmetic for the chain cascades, the preorder sweep on the same
    tiles for alt_tree's stage tree.  Stage taps of two frames, raw candidates (minNeighbors 0) and grouped boxes of all.
    Both prep routes: the four separate kernels (cvResize pyramid, two-pass integrals, k_cv24_variance) and the fused
    k_pyramid_integral<.., SEM = 1> that bench.py's batches take by policy (cvResize arithmetic, integrals and f64
    factors in one march); with taps on, the fused kernel also leaves the pyramid bytes and both integrals to compare."""
    # alt_tree's stage tree (a spine of five stages, then two chains of 18 and 21 stages, the second tried when the first
    # fails) runs as ALTERNATIVE CHAINS on the chain kernel by default -- segment after segment over the same tile, failures
    # of a chain stage collected for the next chain -- and as the preorder sweep (k_cv24_tree) with HC_CV24_SEG=0: the
    # route "fused_sweep", which only exists for that cascade.
    if route == "fused_sweep" and name != "alt_tree":
        pytest.skip("the other cascades' stages form a chain")
    monkeypatch.setenv("HC_FUSED_PREP", "0" if route == "separate" else "1")
    monkeypatch.setenv("HC_CV24_SEG", "0" if route == "fused_sweep" else "1")
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    kinds = ["planted", "texture", "noise"]
    frames = np.stack([synth.make_frames(kinds[i % 3], 1, 240, 320, first=500 + i)[0] for i in range(6)])
    want = [oc.detect_cv24(fr, 1.1, 2, (30, 30), want_taps=True) for fr in frames]
    ctx = pkg.Context(casc)
    ctx.set_semantics(True)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=6, cand_capacity=1 << 14)
    assert [(L.w, L.h, L.ystep, L.nx, L.ny) for L in ctx.levels()] == \
        [(l["w"], l["h"], l["ystep"], l["nx"], l["ny"]) for l in oc.plan_cv24(240, 320, 1.1, (30, 30))]
    ctx.enable_stage_taps(True)
    ctx.enable_timing(True)
    got = ctx.detect_host(frames)
    ms = ctx.kernel_ms()
    # (an interval without a kernel still measures its event records, ~5-10 us: compare with the kernel that did run)
    if route != "separate":
        assert ms["k_integral_rows"] + ms["k_integral_cols"] < 0.5 * ms["k_pyramid_integral"], ms
    else:
        assert ms["k_integral_rows"] + ms["k_integral_cols"] > 0.5 * ms["k_pyramid_integral"], ms
    raws = ctx.raw_candidates(6)
    nraw = 0
    info = ctx.last_launch()
    assert info["prep_route"] == ("cv24_parallel" if route == "separate" else "cv24_fused"), info
    # compiled early stages with the 2.4 node arithmetic (csrc/spec/spec_*24.hip) on the chain kernel -- chains of alt_tree
    # included, whose first six stages keep their file order; the preorder sweep evaluates every stage generically
    assert (ctx.compiled_stages() > 0) == (route != "fused_sweep"), (name, route, ctx.compiled_stages())
    assert info["variant"][0] == (("cv24_tree" if route == "fused_sweep" else "cv24_chains") if name == "alt_tree" else "cv24"), info
    # the f32 variance-factor tap of the >= 4 layout does not exist in this mode (f64 factors live in another buffer): the
    # read must fail cleanly instead of copying from a one-element stub
    with pytest.raises(pkg.HipCascadeError) as ei:
        ctx.read_tap(0, 0, pkg.TAP_VNF)
    assert ei.value.code == -4, ei.value
    for li in (0, len(ctx.levels()) // 2):                  # pyramid bytes and integrals of one frame, two levels
        L = ctx.levels()[li]
        img = oracle.resize_linear_cv24(frames[4], L.h, L.w)
        assert np.array_equal(ctx.read_tap(4, li, pkg.TAP_PYRAMID), img), (name, route, li)
        s_, q_ = oracle.integral(img)
        assert np.array_equal(ctx.read_tap(4, li, pkg.TAP_SUM), s_) and np.array_equal(ctx.read_tap(4, li, pkg.TAP_SQSUM), q_), (name, route, li)
    for f in range(6):
        grouped, raw, otaps = want[f]
        if f in (0, 4):
            for li in range(len(ctx.levels())):
                assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (name, f, li)
        assert np.array_equal(raws[f], raw), (name, f)
        assert np.array_equal(got[f], grouped), (name, f)
        nraw += len(raw)
    assert nraw > 0
    # the production instantiation (no taps), device-resident frames, then back to the >= 4 semantics on the same context
    ctx.enable_stage_taps(False)
    import torch
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ctx.enqueue_device(d.data_ptr(), 6)
    got2 = ctx.wait(6)
    for f in range(6):
        assert np.array_equal(got2[f], want[f][0]), (name, f)
    ctx.set_semantics(False)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=6)
    got4 = ctx.detect_host(frames)
    for f in range(6):
        assert np.array_equal(got4[f], oc.detect(frames[f], 1.1, 2, (30, 30))[0]), (name, f)


@pytest.mark.parametrize("spec", ["1", "0"])
def test_cv24_fused_prep_720p_all_taps(pkg, oracle, synth, monkeypatch, spec):
    """The fused 2.4 prep at the bench geometry: two 1280x720 frames, every level's stage taps (each depends on that window's
    f64 variance factor -- two levels have more than 512 grid columns, whose ring words live in LDS), the pyramid bytes and
    integrals of three levels, raw candidates and boxes."""
    monkeypatch.setenv("HC_FUSED_PREP", "1")
    monkeypatch.setenv("HC_SPEC", spec)          # with the compiled 2.4 stages and with the generic kernel alone
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frames = np.stack([synth.make_frames(k, 1, 720, 1280, first=620 + i)[0] for i, k in enumerate(["planted", "texture"])])
    want = [oc.detect_cv24(fr, 1.1, 2, (30, 30), want_taps=True) for fr in frames]
    ctx = pkg.Context(casc)
    ctx.set_semantics(True)
    ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=2, cand_capacity=1 << 15)
    assert max(L.nx for L in ctx.levels()) > 512
    ctx.enable_stage_taps(True)
    ctx.enable_timing(True)
    got = ctx.detect_host(frames)
    assert (ctx.compiled_stages() > 0) == (spec == "1")
    ms = ctx.kernel_ms()
    assert ms["k_pyramid_integral"] > 0.05 and ms["k_integral_rows"] < 0.2 * ms["k_pyramid_integral"] and \
        ms["k_variance"] < 0.2 * ms["k_pyramid_integral"], ms
    raws = ctx.raw_candidates(2)
    wide = [li for li, L in enumerate(ctx.levels()) if L.nx > 512]
    for li in (0, wide[0], len(ctx.levels()) - 1):
        L = ctx.levels()[li]
        img = oracle.resize_linear_cv24(frames[1], L.h, L.w)
        assert np.array_equal(ctx.read_tap(1, li, pkg.TAP_PYRAMID), img), li
        s_, q_ = oracle.integral(img)
        assert np.array_equal(ctx.read_tap(1, li, pkg.TAP_SUM), s_) and np.array_equal(ctx.read_tap(1, li, pkg.TAP_SQSUM), q_), li
    for f in range(2):
        grouped, raw, otaps = want[f]
        for li in range(len(ctx.levels())):
            st = ctx.read_tap(f, li, pkg.TAP_STAGES)
            assert np.array_equal(st, otaps[li]), "frame %d level %d: %d taps differ" % (f, li, int((st != otaps[li]).sum()))
        assert np.array_equal(raws[f], raw) and np.array_equal(got[f], grouped), f
    assert sum(len(w[0]) for w in want) > 0


def test_cv24_bench_route_720p(pkg, oracle, synth):
    """bench.py --semantics cv24 as a test: 720p, alt, a batch of 24 through the tiled kernel; rank 0 checks frames against the
    oracle's 2.4 restatement before it times anything."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--semantics", "cv24", "--batch", "24", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["parity"]["bit_exact"] is True and out["parity"]["frames_checked"] == 16 and out["parity"]["boxes"] > 0
    assert "2.4" in out["metric"] and out["value"] > 0 and out["roofline"]["kernel"] == "k_cascade"
