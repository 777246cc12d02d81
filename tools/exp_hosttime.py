import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, importlib
import bench
pkg=bench.load_package(); synth=importlib.import_module("ugproj_detector_amd.synth"); shard=importlib.import_module("ugproj_detector_amd.shard")
B=int(sys.argv[1]) if len(sys.argv) > 1 else 64
casc=pkg.Cascade(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "ugproj-detector_amd", "data", "cascades", "haarcascade_frontalface_alt.hcc"))
fr=synth.make_frames('planted',16,720,1280); fr=fr[np.arange(B)%16]
d=torch.from_numpy(fr).cuda(); torch.cuda.synchronize()
ctx=pkg.Context(casc,0,torch.cuda.current_stream().cuda_stream); ctx.configure(720,1280,1.1,2,(30,30),(0,0),B,4096)
for it in range(3):
    t0=time.perf_counter(); ctx.enqueue_device(d.data_ptr(),B); t1=time.perf_counter()
    torch.cuda.synchronize(); t2=time.perf_counter()
    boxes=ctx.wait(B); t3=time.perf_counter()
    rec=shard.pack_records(range(B),boxes); t4=time.perf_counter()
    print("enqueue %.2f ms, gpu %.2f ms, wait(after sync) %.2f ms, pack %.2f ms"%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3))
print('frames grouped on the host:', ctx.group_host_frames()); raw=ctx.raw_candidates(B); print('raw per frame', [len(r) for r in raw][:8], sum(len(r) for r in raw))
