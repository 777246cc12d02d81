"""Wall time of small host batches (1..16 frames of 1280x720) for the small-batch policies:
   python tools/exp_small_batch.py   (set HC_SEG_TILES / HC_SEG_MAX_BATCH / HC_SIDE_MAX_BATCH / HC_FUSED_PREP in the environment)"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench, importlib
pkg = bench.load_package(); synth = importlib.import_module("ugproj_detector_amd.synth")
casc = pkg.Cascade(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "ugproj-detector_amd", "data", "cascades", "haarcascade_frontalface_alt.hcc"))
frames = synth.make_frames("planted", 16, 720, 1280)
ctx = pkg.Context(casc)
ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), 16, 1 << 14)
out = []
for n in (1, 2, 4, 8, 16):
    ctx.detect_host(frames[:n])
    t = []
    for i in range(15):
        t0 = time.perf_counter(); ctx.detect_host(frames[:n]); t.append(time.perf_counter() - t0)
    out.append("%d: %.3f ms" % (n, 1e3 * np.median(t)))
print({k: v for k, v in os.environ.items() if k.startswith("HC_")}, " | ".join(out))
