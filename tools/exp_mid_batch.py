"""Wall time of mid-size host batches (8..128 frames of 1280x720; one context, no pipelining) and the rate they give:
   python tools/exp_mid_batch.py   (HC_SEG_MAX_BATCH / HC_SEG_TILES / HC_FUSED_MIN_BATCH ... in the environment)"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench, importlib
pkg = bench.load_package(); synth = importlib.import_module("ugproj_detector_amd.synth")
casc = pkg.Cascade(pkg.cascade_path("alt"))
frames = synth.make_frames("planted", 32, 720, 1280)
frames = frames[np.arange(128) % 32]
ctx = pkg.Context(casc)
ctx.configure(720, 1280, 1.1, 2, (30, 30), (0, 0), 128, 1 << 13)
ctx.enable_timing(True)
out = []
for n in (8, 16, 24, 32, 48, 64, 96, 128):
    ctx.detect_host(frames[:n])
    t = []
    for i in range(7):
        t0 = time.perf_counter(); ctx.detect_host(frames[:n]); t.append(time.perf_counter() - t0)
    ms = ctx.kernel_ms()
    out.append("%d: %.2f ms = %.0f fps (prep %.2f casc %.2f)" % (n, 1e3 * np.median(t), n / np.median(t), sum(v for k, v in ms.items() if k != "k_cascade"), ms["k_cascade"]))
print({k: v for k, v in os.environ.items() if k.startswith("HC_")}); print("\n".join(out))
