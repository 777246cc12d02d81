"""Single-frame latency of the synchronous drop-in call (hc_detect_multiscale / hc_detect_faces)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import bench, importlib
pkg = bench.load_package(); synth = importlib.import_module("ugproj_detector_amd.synth")
casc = pkg.Cascade(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "ugproj-detector_amd", "data", "cascades", "haarcascade_frontalface_alt.hcc"))
for (rows, cols) in [(360, 640), (720, 1280), (1080, 1920)]:
    fr = synth.planted(rows, cols, synth.frame_seed(0))[0]
    casc.detect_multiscale(fr, 1.1, 2, 0, (30, 30))
    t = []
    for i in range(30):
        t0 = time.perf_counter(); r = casc.detect_multiscale(fr, 1.1, 2, 0, (30, 30)); t.append(time.perf_counter() - t0)
    print("detectMultiScale %dx%d: median %.3f ms, min %.3f ms, %d boxes" % (cols, rows, 1e3 * np.median(t), 1e3 * min(t), len(r)))
bgr = synth.planted_bgr(360, 640, synth.frame_seed(0))[0]
casc.detect_faces(bgr)
t = []
for i in range(30):
    t0 = time.perf_counter(); r = casc.detect_faces(bgr); t.append(time.perf_counter() - t0)
print("detectFaces 640x360 BGR: median %.3f ms, min %.3f ms, %d boxes" % (1e3 * np.median(t), 1e3 * min(t), len(r)))

# where a single 720p frame spends its time: per-kernel HIP-event times of a batch of one
fr = synth.planted(720, 1280, synth.frame_seed(0))[0]
ctx = pkg.Context(casc)
ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), 1, 1 << 14)
ctx.enable_timing(True)
acc = None
for i in range(12):
    t0 = time.perf_counter(); ctx.detect_host(fr[None]); dt = time.perf_counter() - t0
    ms = ctx.kernel_ms()
    if i >= 2:
        acc = dict(ms) if acc is None else {k: acc[k] + v for k, v in ms.items()}
print("720p batch of 1: wall %.3f ms; kernels (ms): %s" % (1e3 * dt, {k: round(v / 10, 3) for k, v in acc.items()}))
