#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: 720p frames/sec, haarcascade_frontalface_alt, boxes bit-exact.

A "step" = one pass of the hot path (pyramid -> integrals -> variance -> cascade -> box
grouping) over one batch of BATCH synthetic 1280x720 gray frames per GPU, frames already
resident in HBM when the timed region starts (BASELINE config #2: alt, scaleFactor 1.1,
minNeighbors 2, minSize 30x30 -- the literal call at reference src/tracker/detector.cc:25-31).
Frames shard across ranks (one process per GPU, weak scaling); the only collective is the
RCCL all-gather of fixed-capacity box records back to rank 0 (ugproj-detector_amd/shard.py).

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against the HBM roof
with the ALGORITHMIC bytes of SURVEY.md 8d (33 813 536 B per 720p frame); `cpu_baseline` is
the CPU oracle (kind "port") timed on this node's host cores on a bounded sample.
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch   # noqa: E402  -- must come before the HIP library: both bind the same libamdhip64.so.7
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ROWS, COLS = 720, 1280
CALL = dict(scale_factor=1.1, min_neighbors=2, min_size=(30, 30), max_size=(0, 0))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def load_package():
    name = "ugproj_detector_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "ugproj-detector_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU per step (one launch of every kernel)")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic frames per rank (cycled)")
    ap.add_argument("--kind", default="planted", choices=["planted", "texture", "noise"])
    ap.add_argument("--cascade", default="alt")
    ap.add_argument("--cpu-sample", type=int, default=32, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="1: both pipeline slots on one stream (clean per-kernel times); 2: a stream per slot")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    pkg = load_package()
    synth = importlib.import_module("ugproj_detector_amd.synth")
    shard = importlib.import_module("ugproj_detector_amd.shard")
    import oracle_lib

    B = args.batch
    cpath = oracle_lib.cascade_path(args.cascade)
    casc = pkg.Cascade(cpath)

    # ---- synthetic frames: rank r owns global frames [r*B, (r+1)*B) of every step's global batch
    ndist = min(args.distinct, B)
    distinct = synth.make_frames(args.kind, ndist, ROWS, COLS, first=rank * ndist)
    frames_np = distinct[np.arange(B) % ndist]
    d_frames = torch.from_numpy(frames_np).to(dev)
    torch.cuda.synchronize()

    # Two contexts (workspaces) on the same stream: batch i+1 is enqueued before batch i is
    # waited for, so the host-side candidate decode / rectangle grouping / record packing of
    # batch i overlaps the kernels of batch i+1.  Every step still does all of its work.
    if args.streams == 1:
        # one explicit (non-default) stream shared by both contexts: kernels of consecutive batches
        # run back to back, never concurrently, so the per-kernel HIP-event times are clean
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
        stream = torch.cuda.current_stream().cuda_stream
    else:
        stream = None   # each context owns a stream: kernels of batch i+1 may overlap the tail of batch i
    ctxs = []
    for _ in range(2):
        c = pkg.Context(casc, device=local_rank, stream=stream)
        c.configure(ROWS, COLS, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], CALL["max_size"],
                    max_batch=B, cand_capacity=4096)
        ctxs.append(c)
    ctx = ctxs[0]
    alg_bytes = ctx.algorithmic_bytes()

    # the RCCL gather of box records runs on its own stream: it must not queue behind the kernels
    # of the next batch that are already enqueued on the compute stream
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None

    def finish(c, global_step):
        boxes = c.wait(B)
        lo = (global_step * world + rank) * B
        rec = shard.pack_records(range(lo, lo + B), boxes)
        if world > 1:
            with torch.cuda.stream(comm_stream):
                allrec = shard.gather_records(rec, dev)
        else:
            allrec = shard.gather_records(rec, None)
        return boxes, allrec

    def run_steps(nsteps, on_done=None):
        """nsteps batches, software-pipelined two deep; returns the results of the last one."""
        res = None
        ctxs[0].enqueue_device(d_frames.data_ptr(), B)
        for i in range(nsteps):
            if i + 1 < nsteps:
                ctxs[(i + 1) & 1].enqueue_device(d_frames.data_ptr(), B)
            res = finish(ctxs[i & 1], i)
            if on_done:
                on_done(ctxs[i & 1])
        return res

    def step(global_step):
        ctx.enqueue_device(d_frames.data_ptr(), B)
        return finish(ctx, global_step)

    run_steps(args.warmup)

    # ---- parity of what is being timed: first `ndist` frames of this rank against the oracle (rank 0 only)
    parity = None
    if rank == 0:
        oc = oracle_lib.OracleCascade(cpath)
        boxes, _ = step(0)
        ncheck = min(4, ndist)
        ok = True
        nb = 0
        for f in range(ncheck):
            want, _ = oc.detect(frames_np[f], CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"])
            ok = ok and np.array_equal(boxes[f], want)
            nb += len(want)
        parity = {"frames_checked": ncheck, "boxes": nb, "bit_exact": bool(ok)}
        if not ok and not os.environ.get("HC_DEBUG_MAX_STAGES"):
            raise SystemExit("bench: HIP boxes differ from the oracle -- refusing to report a number")
    elif world > 1:
        step(0)

    timing = not args.no_timing_events
    for c in ctxs:
        c.enable_timing(timing)
    kernel_ms = {}

    def collect(c):
        if timing:
            for k, v in c.kernel_ms().items():
                kernel_ms[k] = kernel_ms.get(k, 0.0) + v

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, collect)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for c in ctxs:
        c.enable_timing(False)

    if rank == 0:
        fps = world * B * args.steps / elapsed
        out = {
            "metric": "720p frames/sec, haarcascade_frontalface_%s, boxes bit-exact vs CPU oracle" % args.cascade,
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 integrals, f32 features, f64 stage sums", "data": "synthetic",
            "config": {"workload": "BASELINE config #2: 1280x720 gray, haarcascade_frontalface_%s, scaleFactor 1.1, "
                                   "minNeighbors 2, minSize 30x30, S3 '%s' frames" % (args.cascade, args.kind),
                       "frames_per_gpu_per_step": B, "distinct_frames_per_gpu": ndist, "levels": len(ctx.levels()),
                       "parallelism": "frame-shard x%d, RCCL all-gather of box records" % world},
            "parity": parity,
        }
        if timing and kernel_ms:
            per_launch = {k: v / args.steps for k, v in kernel_ms.items()}
            dom = max(per_launch, key=per_launch.get)
            achieved = alg_bytes * B / (per_launch[dom] * 1e-3) / 1e9
            # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
            # command (tools/prof_round.sh -> tools/traffic_from_pmc.py); PMC cannot be collected from
            # inside this process, so the figure is read from profiles/ and is null when absent/stale.
            traffic, traffic_src = None, None
            import glob
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            if tf and args.kind == "planted" and args.cascade == "alt":    # measured at the default workload; scaled per frame
                try:
                    tj = json.load(open(tf[-1]))
                    traffic = int(tj["kernels"][dom]["hbm_bytes_per_launch"] * B / tj.get("frames_per_launch", 64))
                    traffic_src = os.path.relpath(tf[-1], ROOT)
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": alg_bytes * B,
                               "launch_ms": round(per_launch[dom], 4),
                               "all_kernels_ms_per_launch": {k: round(v, 4) for k, v in per_launch.items()}}
        if args.cpu_sample > 0 and world == 1:
            cores = min(os.cpu_count() or 1, 16)   # the GPU box grants a 16-core CPU share per GPU
            ns = args.cpu_sample
            sample = distinct[np.arange(ns) % ndist]
            oc = oracle_lib.OracleCascade(cpath)
            t1 = time.perf_counter()
            oc.detect_many(sample, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], (0, 0), nthreads=cores)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(ns / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": "%d of the same 1280x720 '%s' frames, frame-parallel over %d threads, "
                                             "%.1f s wall" % (ns, args.kind, cores, dt)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
