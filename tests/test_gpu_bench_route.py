"""GPU parity on the route bench.py times, at BASELINE.json's full sizes, and through the shard / gather path.

No environment knobs anywhere in this module: whatever the library's own policy picks for these batch sizes
is what is compared with the oracle -- for B >= 17 frames of 1280x720 that is the fused k_pyramid_integral and
the whole-band `k_cascade<.., SEG = false>` instantiations, i.e. exactly the kernels of BENCH_rNN.json.
"""
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALL = dict(scale_factor=1.1, min_neighbors=2, min_size=(30, 30))      # reference detector.cc:25-31


@pytest.fixture(autouse=True)
def _no_knobs(monkeypatch):
    for k in list(os.environ):
        if k.startswith("HC_"):
            monkeypatch.delenv(k)


def _oracle_many(oc, frames, want_taps):
    """oracle results of every frame, frame-parallel (ctypes releases the GIL)"""
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda f: oc.detect(f, 1.1, 2, (30, 30), (0, 0), want_taps=want_taps), frames))


def test_bench_route_taps_720p_batch24(pkg, oracle, synth):
    """BASELINE config #2 as bench.py runs it: 24 DISTINCT 1280x720 frames (S3 planted, S2 texture, S1 noise
    mixed) in one batch through the default policy.  Per-window stage-pass counts of every level, raw candidates
    in canonical order and grouped boxes (order included) of every frame against the oracle."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    B = 24
    kinds = ["planted", "texture", "noise"]
    frames = np.stack([synth.make_frames(kinds[i % 3], 1, 720, 1280, first=100 + i)[0] for i in range(B)])
    ctx = pkg.Context(casc)
    ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 13)
    ctx.enable_stage_taps(True)
    ctx.enable_timing(True)
    got = ctx.detect_host(frames)
    ms = ctx.kernel_ms()
    assert ms["k_pyramid_integral"] > 0.05 and ms["k_integral_rows"] < 0.03 and ms["k_integral_cols"] < 0.03, \
        "a batch of 24 720p frames must take the fused prep kernel by policy: %r" % ms
    raws = ctx.raw_candidates(B)
    want = _oracle_many(oc, frames, True)
    nlev = len(ctx.levels())
    assert nlev == 33
    nboxes = 0
    for f in range(B):
        grouped, raw, otaps = want[f]
        for li in range(nlev):
            st = ctx.read_tap(f, li, pkg.TAP_STAGES)
            assert np.array_equal(st, otaps[li]), "frame %d level %d: %d stage counts differ" % (f, li, int((st != otaps[li]).sum()))
        assert np.array_equal(raws[f], raw), "frame %d raw candidates" % f
        assert np.array_equal(got[f], grouped), "frame %d boxes" % f
        nboxes += len(grouped)
    assert nboxes >= 20
    # the integrals of one mid-batch frame, largest and a middle level
    for li in (0, 7):
        L = ctx.levels()[li]
        img = oracle.resize_linear_exact(frames[13], L.h, L.w)
        s, q = oracle.integral(img)
        assert np.array_equal(ctx.read_tap(13, li, pkg.TAP_SUM), s) and np.array_equal(ctx.read_tap(13, li, pkg.TAP_SQSUM), q)
    # The instantiation BENCH_rNN.json times has the stage taps compiled OUT (k_cascade<.., TAPS = false, ..>): the same
    # 24 frames again without taps, raw candidates (canonical order) and grouped boxes of every frame.
    ctx.enable_stage_taps(False)
    assert ctx.compiled_stages() > 0, "the stock cascade must run its compiled early stages on this route"
    got2 = ctx.detect_host(frames)
    assert ctx.last_launch()["device_grouping"] == 1 and ctx.group_host_frames() == 0, \
        "a batch of 24 frames groups its rectangles on the device (K6) by policy"
    raws2 = ctx.raw_candidates(B)
    for f in range(B):
        assert np.array_equal(raws2[f], want[f][1]), "no-tap kernel: frame %d raw candidates" % f
        assert np.array_equal(got2[f], want[f][0]), "no-tap kernel: frame %d boxes" % f
    # and from device-resident frames through the asynchronous enqueue / wait pair, as bench.py feeds it
    import torch
    d = torch.from_numpy(frames).cuda()
    ctx.enqueue_device(d.data_ptr(), B)
    got3 = ctx.wait(B)
    raws3 = ctx.raw_candidates(B)
    for f in range(B):
        assert np.array_equal(raws3[f], want[f][1]) and np.array_equal(got3[f], want[f][0]), "device input: frame %d" % f


def test_bench_shape_160_frames_every_chunk(pkg, oracle, synth):
    """The code path only big batches take, at the real shape and with the library's own knobs: 160 DISTINCT 1280x720
    frames in one batch are five k_pyramid_integral chunks of 32 frames and, for k_cascade, one full chunk of 128 plus a
    short one of 32 -- both workgroup remaps, including the tail branch of the cascade kernel's.  Raw candidates (canonical
    order) and grouped boxes of ALL 160 frames against the oracle; the launch report proves the remaps ran with those
    values and which kernel builds evaluated the frames (every frame's boxes are consumed downstream: reference
    src/tracker/face_tracker.cc:290-304)."""
    import torch
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    B = 160
    kinds = ["planted", "planted", "texture", "noise"]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        frames = np.stack(list(ex.map(lambda i: synth.make_frames(kinds[i % 4], 1, 720, 1280, first=3000 + i)[0], range(B))))
    ctx = pkg.Context(casc)
    ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 13)
    d = torch.from_numpy(frames).cuda()
    ctx.enqueue_device(d.data_ptr(), B)
    got = ctx.wait(B)
    raws = ctx.raw_candidates(B)
    info = ctx.last_launch()
    assert info["prep_route"] == "fused" and info["variance_route"] == "ring_strips" and info["nframes"] == B, info
    assert (info["pi_frame_chunk"], info["pi_remapped"], info["pi_tail_frames"]) == (32, 1, 0), info
    assert (info["casc_frame_chunk"], info["casc_remapped"], info["casc_tail_frames"]) == (128, 1, 32), info
    assert info["variant"] == ("p16", "spec") and info["segments"] == 0 and info["sq_stored"] == 0, info
    assert info["device_grouping"] == 1 and ctx.group_host_frames() == 0, "the bench shape groups on the device (K6)"
    want = _oracle_many(oc, frames, False)
    nboxes = 0
    for f in range(B):
        assert np.array_equal(raws[f], want[f][1]), "frame %d raw candidates" % f
        assert np.array_equal(got[f], want[f][0]), "frame %d boxes" % f
        nboxes += len(want[f][0])
    assert nboxes >= B          # the planted half alone carries several faces per frame


@pytest.mark.parametrize("name", ["alt2", "default", "alt_tree"])
def test_full_size_2160p_other_cascades(pkg, oracle, synth, name):
    """BASELINE config #5 (all four cascades at 3840x2160; `alt` is covered in test_gpu_parity.py): raw candidates
    in canonical order and grouped boxes of one planted 4K frame."""
    path = oracle.cascade_path(name)
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    frame = synth.planted(2160, 3840, synth.frame_seed(3), nfaces=8, min_size=60, max_size=500)[0]
    ctx = pkg.Context(casc)
    ctx.configure(2160, 3840, 1.1, 2, (30, 30), (0, 0), max_batch=1, cand_capacity=1 << 16)
    assert len(ctx.levels()) == 45
    assert ctx.algorithmic_bytes() == (440608352 if name == "default" else 303677664)
    got = ctx.detect_host(frame[None])
    grouped, raw = oc.detect(frame, 1.1, 2, (30, 30), (0, 0))
    assert np.array_equal(ctx.raw_candidates(1)[0], raw)
    assert np.array_equal(got[0], grouped) and len(grouped) >= 4


def test_shared_prep_four_cascades_2160p(pkg, oracle, synth):
    """BASELINE config #5 as ONE pass at its full size: the four cascades of data/ over one planted 3840x2160 frame
    through hc_context_configure_shared (47-level union ladder; the prep route is whatever the cost model picks for
    one 4K frame).  Raw candidates in canonical order and grouped boxes of every member against the oracle."""
    names = ("alt", "alt2", "alt_tree", "default")
    frame = synth.planted(2160, 3840, synth.frame_seed(5), nfaces=8, min_size=60, max_size=500)[0]
    cascs = [pkg.Cascade(oracle.cascade_path(n)) for n in names]
    ocs = [oracle.OracleCascade(oracle.cascade_path(n)) for n in names]
    ctxs = [pkg.Context(c) for c in cascs]
    pkg.configure_shared(ctxs, 2160, 3840, 1.1, 2, (30, 30), (0, 0), max_batch=1, cand_capacity=1 << 16)
    assert len(ctxs[0].levels()) == 47
    assert [sum(1 for L in c.levels() if L.nx > 0) for c in ctxs] == [45, 45, 45, 45]
    assert [c.algorithmic_bytes() for c in ctxs] == [303677664, 303677664, 303677664, 440608352]
    with ThreadPoolExecutor(max_workers=4) as ex:
        want = list(ex.map(lambda oc: oc.detect(frame, 1.1, 2, (30, 30), (0, 0)), ocs))
    for c in ctxs:
        c.enable_timing(True)
    for c in ctxs:                                            # leader first, then every follower once
        c.enqueue_host(frame[None])
    total = 0
    var_ms = []
    for c, (grouped, raw), name in zip(ctxs, want, names):
        got = c.wait(1)
        assert np.array_equal(c.raw_candidates(1)[0], raw), name
        assert np.array_equal(got[0], grouped), name
        total += len(grouped)
        var_ms.append(c.kernel_ms()["k_variance"])
    assert total >= 16
    # alt2 and alt_tree have the leader's 20 x 20 window: they scan with ITS variance factors (no k_variance of their
    # own); default (24 x 24) computes its own
    # (an interval without a kernel still measures the event records themselves: ~10 us)
    assert var_ms[0] > 0.04 and var_ms[3] > 0.04 and var_ms[1] < 0.4 * var_ms[0] and var_ms[2] < 0.4 * var_ms[0], var_ms
    # the batch again: the shared factors are rewritten by the leader while the followers' last scans are long done
    frame2 = synth.planted(2160, 3840, synth.frame_seed(6), nfaces=6, min_size=60, max_size=400)[0]
    for c in ctxs:
        c.enqueue_host(frame2[None])
    for c, oc, name in zip(ctxs, ocs, names):
        got = c.wait(1)
        if name in ("alt2", "alt_tree"):
            assert np.array_equal(got[0], oc.detect(frame2, 1.1, 2, (30, 30), (0, 0))[0]), name
    for c in ctxs:
        c.close()


def test_wait_does_not_wait_for_the_next_slot(pkg, oracle, synth):
    """Two pipeline slots of a two-cascade shared pass = four contexts on one stream (bench.py's layout for BASELINE
    config #3).  The wait for slot A's results must return when A's kernels are done, not when slot B's batch -- enqueued
    behind it -- has drained as well: the runtime maps the streams of four contexts onto shared hardware queues, where a
    stream / event wait returns a whole batch late, so results are published to host memory by the batch's own last
    kernel and polled for (api_wait.cpp wait_published).  Same frames in both slots, so B takes as long as A."""
    import time
    import torch
    B = 24
    frames = np.stack([synth.make_frames("planted", 1, 720, 1280, first=700 + i)[0] for i in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    stream = torch.cuda.Stream()
    cascs = [pkg.Cascade(oracle.cascade_path(n)) for n in ("alt2", "default")]
    slots = []
    for _ in range(2):
        cs = [pkg.Context(c, device=0, stream=stream.cuda_stream) for c in cascs]
        pkg.configure_shared(cs, 720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 14)
        slots.append(cs)

    def enqueue(slot):
        for c in slot:
            c.enqueue_device(d_frames.data_ptr(), B)

    first = None
    b_still_running = []
    for rep in range(3):                                      # first repetition: allocations, code objects
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueue(slots[0]); enqueue(slots[1])
        done_b = torch.cuda.Event()
        done_b.record(stream)                                 # behind slot B's kernels on the shared stream
        got_a = [c.wait(B) for c in slots[0]]
        # the evidence is the device's own: when slot A's results are back, slot B's batch (as long as A's: >= 10 ms of
        # kernels) must still be running.  A wait that went through the runtime's queue drain returned only after it.
        b_still_running.append(not done_b.query())
        t1 = time.perf_counter()
        got_b = [c.wait(B) for c in slots[1]]
        t2 = time.perf_counter()
        if first is None:
            first = got_a
        for x, y in zip(got_a + got_b, first + first):
            assert all(np.array_equal(p, q) for p, q in zip(x, y))
    assert any(b_still_running[1:]), "slot A's wait returned only when slot B was done (%.2f ms of %.2f ms): %r" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t0), b_still_running)
    oc = oracle.OracleCascade(oracle.cascade_path("alt2"))
    assert np.array_equal(first[0][3], oc.detect(frames[3], 1.1, 2, (30, 30), (0, 0))[0])
    for cs in slots:
        for c in cs:
            c.close()


def test_1080p_batch_two_cascades(pkg, oracle, synth):
    """BASELINE config #3 with a batch > 1: 8 distinct 1920x1080 frames (18 720p-equivalents: fused prep and whole
    bands by policy) through alt2 and default; stage counts of two frames, raw candidates and boxes of all."""
    frames = np.stack([synth.make_frames(["planted", "texture"][i % 2], 1, 1080, 1920, first=300 + i)[0] for i in range(8)])
    for name in ("alt2", "default"):
        path = oracle.cascade_path(name)
        casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
        ctx = pkg.Context(casc)
        ctx.configure(1080, 1920, 1.1, 2, (30, 30), (0, 0), max_batch=8, cand_capacity=1 << 14)
        ctx.enable_stage_taps(True)
        got = ctx.detect_host(frames)
        raws = ctx.raw_candidates(8)
        want = _oracle_many(oc, frames, True)
        for f in range(8):
            grouped, raw, otaps = want[f]
            if f in (2, 5):
                for li in range(len(ctx.levels())):
                    assert np.array_equal(ctx.read_tap(f, li, pkg.TAP_STAGES), otaps[li]), (name, f, li)
            assert np.array_equal(raws[f], raw), (name, f)
            assert np.array_equal(got[f], grouped), (name, f)
        ctx.close()


def test_one_shard_equals_two_shards(pkg, oracle, synth):
    """the same frames through 1 vs N shards give identical per-frame lists.  2*B frames as ONE
    batch of one context, and as two shards (two contexts, B frames each) whose records go through the gather's
    pack / overflow / unpack path.  minNeighbors 0 returns the raw candidates as boxes, so several frames carry
    far more than shard.MAX_BOXES rectangles and the overflow block is exercised with real detections."""
    import importlib
    shard = importlib.import_module("ugproj_detector_amd.shard")
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    B = 6
    frames = np.stack([synth.planted(240, 320, synth.frame_seed(40 + i), nfaces=3, min_size=40, max_size=110)[0]
                       for i in range(2 * B)])
    for mn in (2, 0):
        one = pkg.Context(casc)
        one.configure(240, 320, 1.1, mn, (30, 30), (0, 0), max_batch=2 * B, cand_capacity=1 << 14)
        whole = one.detect_host(frames)
        merged = {}
        nover = 0
        for r in range(2):
            c = pkg.Context(casc)
            c.configure(240, 320, 1.1, mn, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 14)
            boxes = c.detect_host(frames[r * B:(r + 1) * B])
            ids = list(range(r * B, (r + 1) * B))
            rec = shard.pack_records(ids, boxes)
            ov = shard.pack_overflow(boxes)
            nover += len(ov)
            assert shard.overflow_counts(rec, 1)[0] == len(ov)
            for fid, (cnt, b) in shard.unpack_records(rec, [ov]).items():
                merged[fid] = b
            c.close()
        assert sorted(merged) == list(range(2 * B))
        for f in range(2 * B):
            assert np.array_equal(merged[f], whole[f]), (mn, f)
            want, _ = oc.detect(frames[f], 1.1, mn, (30, 30))
            assert np.array_equal(whole[f], want), (mn, f)
        if mn == 0:
            assert nover > 0 and max(len(b) for b in whole) > shard.MAX_BOXES, "stimulus no longer overflows a record"
        one.close()


def _run_bench(extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(js) == 1, r.stdout[-2000:]
    return json.loads(js[0])


def test_bench_two_ranks_on_one_gpu(pkg):
    """`python bench.py --gpus 2` as the driver invokes it (no launcher, no WORLD_SIZE): bench.py spawns its own
    ranks.  This box has one GPU, so the two ranks share it and the gather runs over gloo (RCCL refuses two ranks
    on one device); everything else -- frame sharding, record packing, the gather, rank 0 checking frames of BOTH
    ranks against the oracle before timing -- is the path of the 8-GPU run."""
    out = _run_bench(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "3", "--warmup", "1", "--batch", "20",
                      "--cpu-sample", "0"])
    assert out["parity"]["bit_exact"] is True
    assert out["parity"]["ranks_checked"] == 2 and out["parity"]["gathered_frames"] == 40
    assert out["parity"]["frames_checked"] == 16
    assert out["value"] > 0 and out["config"]["frames_per_gpu_per_step"] == 20
    # two ranks on one device is a rehearsal of the code path, not a scaling point: the line says so
    assert out["n_gpus"] == 1 and out["ranks"] == 2 and out["rehearsal"] is True and "roofline" not in out
    assert len(out["per_rank_fps"]) == 2 and "cpu_baseline" not in out


def test_bench_five_ranks_rehearsal(pkg):
    """The N = 8 code path with as many ranks as a one-GPU box admits: its process guard allows six processes on
    the card, this pytest process is one of them, so five self-spawned ranks share the GPU (gloo transport).  Every
    rank's frames come back through the gather and rank 0 checks frames of EVERY rank against the oracle."""
    out = _run_bench(["--gpus", "5", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1", "--batch", "16",
                      "--cpu-sample", "4"])
    assert out["parity"]["bit_exact"] is True and out["parity"]["ranks_checked"] == 5
    assert out["parity"]["gathered_frames"] == 80 and out["parity"]["frames_checked"] == 15
    assert out["ranks"] == 5 and out["rehearsal"] is True and len(out["per_rank_fps"]) == 5
    assert out["cpu_baseline"]["value"] > 0            # reported by rank 0 at any N


def test_bench_rccl_gather_single_rank(pkg):
    """The RCCL transport itself on hardware: one rank, process group `nccl`, the record all-gather on its own
    communication stream with pinned staging (world size 1 is what a one-GPU box can give RCCL)."""
    out = _run_bench(["--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1", "--batch", "20", "--cpu-sample", "0"])
    assert out["parity"]["bit_exact"] is True and out["parity"]["gathered_frames"] == 20
    assert "RCCL" in out["config"]["parallelism"] and "roofline" in out and out["n_gpus"] == 1 and "rehearsal" not in out


def test_bench_default_line_has_extra_configs(pkg):
    """The driver's command (`python bench.py`, here with fewer steps): config #2 is `value`; configs #3 and #5 follow
    outside its timed region under extra.configs, each with its own oracle check and roofline fraction; no process
    group at N = 1 and the line says so."""
    out = _run_bench(["--steps", "4", "--warmup", "1", "--extra-steps", "2", "--cpu-sample", "16"], timeout=1100)
    assert out["n_gpus"] == 1 and out["parity"]["bit_exact"] is True and out["parity"]["frames_checked"] == 16
    assert "no collective" in out["config"]["parallelism"] and "720p" in out["metric"]
    assert out["roofline"]["kernel"] == "k_cascade" and 0 < out["roofline"]["frac"] < 1
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    ex = out["extra"]["configs"]
    assert sorted(ex) == ["3", "5"]
    for k, nf in (("3", 384), ("5", 128)):
        assert "error" not in ex[k], ex[k]
        assert ex[k]["parity"]["bit_exact"] is True and ex[k]["frames_per_step"] == nf
        assert ex[k]["value"] > 0 and 0 < ex[k]["roofline"]["frac"] < 1


@pytest.mark.parametrize("min_neighbors", [2, 0])
def test_cpp_multi_gpu_demo_single_device(pkg, oracle, synth, tmp_path, min_neighbors):
    """SURVEY section 7 step 10 from plain C++ (hip_cascade/multi_gpu_demo.cpp): one process, ncclCommInitAll, a feeder
    thread and a context pair per GPU, grouped ncclAllGather of box records (+ the overflow round with minNeighbors 0,
    where frames carry hundreds of rectangles), HCBX file written through hc_boxes_save.  One device is what this box
    has; the gathered, frame-ordered box file must equal the oracle frame by frame, ragged last batch included."""
    demo = os.path.join(os.path.dirname(pkg.LIB_PATH), "multi_gpu_demo")
    n = 11                                                   # batch 4: two full global batches + a ragged one
    frames = np.stack([synth.planted(240, 320, synth.frame_seed(70 + i), nfaces=3, min_size=40, max_size=110)[0] for i in range(n)])
    f = tmp_path / "frames.u8"
    frames.tofile(f)
    out = tmp_path / "boxes.hcbx"
    r = subprocess.run([demo, oracle.cascade_path("alt"), str(f), str(n), "240", "320", "1", "4", "2", str(out), str(min_neighbors)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    got = pkg.load_boxes(str(out))
    assert sorted(got) == list(range(n))
    oc = oracle.OracleCascade(oracle.cascade_path("alt"))
    total = 0
    for i in range(n):
        want, _ = oc.detect(frames[i], 1.1, min_neighbors, (30, 30))
        assert np.array_equal(got[i], want), i
        total += len(want)
    assert ("%d boxes per pass" % total) in r.stdout, r.stdout
    if min_neighbors == 0:
        assert max(len(b) for b in got.values()) > 64


def _scanned_level_taps(ctx, pkg, frame_index):
    """stage taps of the levels a member of a shared group scans itself, in ladder order"""
    return [ctx.read_tap(frame_index, li, pkg.TAP_STAGES) for li, L in enumerate(ctx.levels()) if L.nx > 0]


@pytest.mark.parametrize("names,shape", [(("alt2", "default"), (360, 640)), (("default", "alt", "alt_tree", "alt2"), (240, 320)),
                                         (("alt2", "default"), (1080, 1920))])
def test_shared_prep_pass_equals_separate_contexts(pkg, oracle, synth, names, shape):
    """BASELINE configs #3 / #5 as ONE pass: contexts of several cascades configured with hc_context_configure_shared
    take their integrals from one pyramid + integral pass over the union ladder (win 20 and win 24 ladders differ at
    both ends).  Each member's stage taps, raw candidates and boxes equal the oracle's for its cascade -- two
    consecutive batches, leader first then the followers, every member on its own stream."""
    rows, cols = shape
    nf = 3 if rows < 1000 else 2
    batches = [np.stack([synth.planted(rows, cols, synth.frame_seed(80 + 10 * b + i), nfaces=4, min_size=40,
                                       max_size=min(rows, cols) // 2)[0] for i in range(nf)]) for b in range(2)]
    cascs = [pkg.Cascade(oracle.cascade_path(n)) for n in names]
    ocs = [oracle.OracleCascade(oracle.cascade_path(n)) for n in names]
    ctxs = [pkg.Context(c) for c in cascs]
    pkg.configure_shared(ctxs, rows, cols, 1.1, 2, (30, 30), (0, 0), max_batch=nf, cand_capacity=1 << 14)
    nlev = len(ctxs[0].levels())
    assert all(len(c.levels()) == nlev for c in ctxs)
    own = [len(oc.plan(rows, cols, 1.1, (30, 30))[0]) for oc in ocs]
    assert nlev >= max(own) and [sum(1 for L in c.levels() if L.nx > 0) for c in ctxs] == own
    if set(names) == {"alt2", "default"} and rows == 1080:
        assert nlev == 39 and own == [37, 37]
        assert [c.algorithmic_bytes() for c in ctxs] == [76051696, 110293312]        # each cascade's own full figure
    for c in ctxs:
        c.enable_stage_taps(True)
    with pytest.raises(pkg.HipCascadeError):
        ctxs[1].enqueue_host(batches[0])                      # a follower cannot run ahead of its leader
    for frames in batches:
        for c in ctxs:                                        # leader first, then every follower once
            c.enqueue_host(frames)
        for c, oc, name in zip(ctxs, ocs, names):
            got = c.wait(nf)
            raws = c.raw_candidates(nf)
            for f in range(nf):
                grouped, raw, otaps = oc.detect(frames[f], 1.1, 2, (30, 30), (0, 0), want_taps=True)
                taps = _scanned_level_taps(c, pkg, f)
                assert len(taps) == len(otaps)
                for li, (a, b) in enumerate(zip(taps, otaps)):
                    assert np.array_equal(a, b), (name, f, li)
                assert np.array_equal(raws[f], raw), (name, f)
                assert np.array_equal(got[f], grouped), (name, f)
    # the leader alone (followers skipped for a batch) still works, and a reconfigured leader detaches its followers
    ctxs[0].enqueue_host(batches[0])
    first = ctxs[0].wait(nf)
    want0 = [ocs[0].detect(fr, 1.1, 2, (30, 30))[0] for fr in batches[0]]
    assert all(np.array_equal(a, b) for a, b in zip(first, want0))
    ctxs[0].configure(rows, cols, 1.1, 2, (30, 30), (0, 0), max_batch=nf)
    with pytest.raises(pkg.HipCascadeError):
        ctxs[1].enqueue_host(batches[0])


def test_cpp_drivers_make_their_own_frames(pkg, oracle, synth, tmp_path):
    """throughput_demo and multi_gpu_demo with `synth:planted:<first>` instead of a frame file: the C++ generator
    (hip_cascade/synth.hpp) feeds the batched C ABI, and what comes back equals the oracle on the frames the Python
    generator makes for the same indices."""
    libdir = os.path.dirname(pkg.LIB_PATH)
    n, first = 7, 40
    frames = synth.make_frames("planted", n, 240, 320, first=first)
    oc = oracle.OracleCascade(oracle.cascade_path("alt"))
    want = [oc.detect(fr, 1.1, 2, (30, 30))[0] for fr in frames]
    total = sum(len(w) for w in want)
    assert total > 0
    out = tmp_path / "boxes.hcbx"
    r = subprocess.run([os.path.join(libdir, "multi_gpu_demo"), oracle.cascade_path("alt"), "synth:planted:%d" % first, str(n), "240", "320",
                        "1", "4", "2", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    got = pkg.load_boxes(str(out))
    for i in range(n):
        assert np.array_equal(got[i], want[i]), i
    r = subprocess.run([os.path.join(libdir, "throughput_demo"), oracle.cascade_path("alt"), "synth:planted:%d" % first, str(n), "240", "320",
                        "4", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert ("%d boxes per pass" % total) in r.stdout, r.stdout


# ---------------------------------------------------------------------------------------------------------------------
# K6: rectangle grouping on the device (group_kernels.hip) -- the default result path of every batch in this file and in
# the parity suites; here it is compared with the host grouping directly and pushed to its limits.

def _key_cloud(rng, levels, nframes, per_frame, centres):
    """raw candidate keys of `nframes` frames: `per_frame[f]` windows scattered around a few face centres over
    neighbouring levels (what a detector's raw output looks like: many similar rectangles per face, some loners)"""
    keys = []
    for f in range(nframes):
        cs = [(int(rng.integers(80, 1100)), int(rng.integers(60, 560)), int(rng.integers(0, len(levels) - 4))) for _ in range(centres)]
        for _ in range(per_frame[f]):
            cx, cy, l0 = cs[int(rng.integers(0, centres))]
            if rng.random() < 0.1:      # a loner somewhere else
                cx, cy = int(rng.integers(0, 1200)), int(rng.integers(0, 640))
            li = min(len(levels) - 1, max(0, l0 + int(rng.integers(-1, 4))))
            L = levels[li]
            px = cx + int(rng.integers(-6, 7)) * max(1, int(L.scale))
            py = cy + int(rng.integers(-6, 7)) * max(1, int(L.scale))
            gx = min(max(int(px / L.scale) // L.ystep, 0), L.nx - 1)
            gy = min(max(int(py / L.scale) // L.ystep, 0), L.ny - 1)
            keys.append((f << 40) | (li << 32) | (gy << 16) | gx)
    keys = np.array(keys, np.uint64)
    rng.shuffle(keys)
    return keys


@pytest.mark.parametrize("cv24", [False, True])
def test_device_grouping_on_key_clouds(pkg, oracle, cv24):
    """The result side alone, fed with synthetic candidate keys: frames with 0, 1, 2, 63, 64, 65, 300, 1000 and 1024 raw
    rectangles (one bin is 1024) in shuffled order.  Device grouping, host grouping and the oracle's groupRectangles on the
    decoded rectangles agree box for box, order included; raw candidates come back in upstream's order on both paths;
    minNeighbors 1, 2 and 5."""
    casc = pkg.Cascade(oracle.cascade_path("alt"))
    rng = np.random.default_rng(77)
    counts = [0, 1, 2, 63, 64, 65, 300, 1000, 1024, 17, 0, 129]
    B = len(counts)
    for mn in (2, 1, 5):
        dev, host = pkg.Context(casc), pkg.Context(casc)
        for c in (dev, host):
            if cv24:
                c.set_semantics(True)
            c.configure(720, 1280, 1.1, mn, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 10)
        dev.set_device_grouping(True)
        host.set_device_grouping(False)
        keys = _key_cloud(rng, dev.levels(), B, counts, centres=5)
        assert len(keys) == sum(counts)
        dev.enqueue_keys(keys, B)
        gd = dev.wait(B)
        assert dev.group_host_frames() == 0 and dev.last_launch()["device_grouping"] == 1
        host.enqueue_keys(keys, B)
        gh = host.wait(B)
        assert host.group_host_frames() == B and host.last_launch()["device_grouping"] == 0
        rd, rh = dev.raw_candidates(B), host.raw_candidates(B)
        nb = 0
        for f in range(B):
            assert len(rd[f]) == counts[f] and np.array_equal(rd[f], rh[f]), "frame %d raw rectangles" % f
            assert np.array_equal(gd[f], gh[f]), "minNeighbors %d frame %d: device and host grouping differ" % (mn, f)
            want = oracle.group_rectangles(rd[f], mn) if len(rd[f]) else np.zeros((0, 4), np.int32)
            assert np.array_equal(gd[f], want), "minNeighbors %d frame %d: device grouping vs the oracle's groupRectangles" % (mn, f)
            nb += len(gd[f])
        assert nb >= 10, "the clouds must form groups (got %d boxes)" % nb


def test_device_grouping_leaves_overflowing_frames_to_the_host(pkg, oracle, synth, monkeypatch):
    """Bins of 8 raw candidates / output slots of 2 boxes (test knobs): the planted frames overflow and are grouped on the
    host from the raw keys, the noise frames stay on the device -- one batch mixes both; boxes and raw candidates of every
    frame equal the oracle's either way, and the counters say which path each frame took."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    B = 12
    kinds = ["planted", "noise", "texture"]
    frames = np.stack([synth.make_frames(kinds[i % 3], 1, 720, 1280, first=300 + i)[0] for i in range(B)])
    want = _oracle_many(oc, frames, False)
    for env, lo, hi in (({"HC_GROUP_BIN_CAP": "8"}, 1, B - 1), ({"HC_GROUP_OUT_CAP": "1"}, 1, B - 1), ({}, 0, 0)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = pkg.Context(casc)
        for k in env:
            monkeypatch.delenv(k)
        ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 13)
        for rnd in range(2):
            got = ctx.detect_host(frames)
            assert lo <= ctx.group_host_frames() <= hi, "%r: %d frames went to the host" % (env, ctx.group_host_frames())
            assert ctx.last_launch()["device_grouping"] == 1
            raws = ctx.raw_candidates(B)
            for f in range(B):
                assert np.array_equal(raws[f], want[f][1]), "%r frame %d raw candidates" % (env, f)
                assert np.array_equal(got[f], want[f][0]), "%r frame %d boxes" % (env, f)
    # the switch off: same boxes, every frame grouped on the host
    ctx.set_device_grouping(False)
    got = ctx.detect_host(frames)
    assert ctx.group_host_frames() == B and ctx.last_launch()["device_grouping"] == 0
    for f in range(B):
        assert np.array_equal(got[f], want[f][0])
    # small batches are latency-bound and grouped on the host by policy (HC_GROUP_MIN_BATCH = 8); the knob moves the bound
    ctx.set_device_grouping(True)
    got = ctx.detect_host(frames[:4])
    assert ctx.group_host_frames() == 4 and ctx.last_launch()["device_grouping"] == 0
    monkeypatch.setenv("HC_GROUP_MIN_BATCH", "1")
    ctx1 = pkg.Context(casc)
    monkeypatch.delenv("HC_GROUP_MIN_BATCH")
    ctx1.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 13)
    got1 = ctx1.detect_host(frames[:1])
    assert ctx1.group_host_frames() == 0 and ctx1.last_launch()["device_grouping"] == 1
    assert np.array_equal(got1[0], want[0][0]) and np.array_equal(got[0], want[0][0])
    # minNeighbors 0: the raw rectangles are the result, nothing to group on either side
    ctx0 = pkg.Context(casc)
    ctx0.configure(720, 1280, 1.1, 0, (30, 30), (0, 0), max_batch=B, cand_capacity=1 << 13)
    got0 = ctx0.detect_host(frames)
    assert ctx0.last_launch()["device_grouping"] == 0
    for f in range(B):
        assert np.array_equal(got0[f], want[f][1])


def test_device_grouping_survives_a_candidate_overflow(pkg, oracle, synth):
    """A batch that groups on the device and overflows its raw-candidate workspace: the first pass's grouping saw a
    truncated list and is discarded with it, the wait grows the workspace and runs the batch -- grouping kernels included --
    again; boxes and raw candidates of all frames equal the oracle's, and the grown workspace serves the next batch at once."""
    path = oracle.cascade_path("alt")
    casc, oc = pkg.Cascade(path), oracle.OracleCascade(path)
    B = 10
    frames = np.stack([synth.planted(240, 320, synth.frame_seed(40 + i), nfaces=3, min_size=40, max_size=110)[0] for i in range(B)])
    ctx = pkg.Context(casc)
    ctx.configure(240, 320, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=4)
    got = ctx.detect_host(frames)
    assert ctx.last_launch()["device_grouping"] == 1 and ctx.group_host_frames() == 0
    raws = ctx.raw_candidates(B)
    again = ctx.detect_host(frames)
    nraw = 0
    for f in range(B):
        want, raw = oc.detect(frames[f], 1.1, 2, (30, 30))
        nraw += len(raw)
        assert np.array_equal(got[f], want) and np.array_equal(raws[f], raw) and np.array_equal(again[f], want), "frame %d" % f
    assert nraw > 4 * B, "the batch must overflow a workspace of %d candidates (it has %d)" % (4 * B, nraw)
