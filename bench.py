#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: 720p frames/sec, haarcascade_frontalface_alt, boxes bit-exact.

A "step" = one pass of the hot path (pyramid -> integrals -> variance -> cascade -> box
grouping) over one batch of synthetic gray frames per GPU, frames already resident in HBM when
the timed region starts.  Workloads (`--config`, BASELINE.json `configs`):
  2 (default, the configuration the metric is quoted on): 1280x720, alt, scaleFactor 1.1,
    minNeighbors 2, minSize 30x30 -- the literal call at reference src/tracker/detector.cc:25-31;
  3: 1920x1080, alt2 + default on the same frames (two-cascade pass);
  5: 3840x2160, all four data/ cascades on the same frames.
Frames shard across ranks (one process per GPU, weak scaling); the only collective is the RCCL
all-gather of box records back to rank 0 (ugproj-detector_amd/shard.py), lossless for any number
of boxes per frame.

Launch: `python bench.py --gpus N` spawns its own N rank processes (the parent never touches
the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the
ranks torchrun started are used as they are.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel against the HBM roof
with the ALGORITHMIC bytes of SURVEY.md 8d (33 813 536 B per 720p/alt frame); `cpu_baseline` is
the CPU oracle (kind "port") timed on this node's host cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CALL = dict(scale_factor=1.1, min_neighbors=2, min_size=(30, 30), max_size=(0, 0))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    2: dict(rows=720, cols=1280, cascades=("alt",), batch=256, cpu_sample=32, parity_frames=4,
            label="BASELINE config #2: 1280x720 gray, haarcascade_frontalface_alt"),
    3: dict(rows=1080, cols=1920, cascades=("alt2", "default"), batch=96, cpu_sample=16, parity_frames=2,
            label="BASELINE config #3: 1920x1080 gray, haarcascade_frontalface_alt2 + _default on the same frames"),
    5: dict(rows=2160, cols=3840, cascades=("alt", "alt2", "alt_tree", "default"), batch=32, cpu_sample=8, parity_frames=1,
            label="BASELINE config #5: 3840x2160 gray, all four data/ cascades on the same frames"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (one launch of every kernel); 0 = the config's")
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic frames per rank (0 = one per batch slot)")
    ap.add_argument("--kind", default="planted", choices=["planted", "texture", "noise"])
    ap.add_argument("--cascade", default="", help="override the config's cascade list (comma separated)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="1: both pipeline slots on one stream (clean per-kernel times); 2: a stream per slot")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="transport of the box gather: nccl = RCCL over xGMI; gloo = rehearsal on a box with fewer GPUs than ranks")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs (rank r -> device r mod count); needs --backend gloo")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the gather even with one rank")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same script
    (fresh interpreters, this parent never initialises HIP and never execs), relay rank 0's JSON
    line, fail if any rank fails."""
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = procs[0].stdout
    lines = []
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad and failed is None:
            failed = bad[0]
            for p in procs:                      # exact PIDs we started, nothing else
                if p.poll() is None:
                    p.terminate()
        if all(c is not None for c in codes):
            break
        time.sleep(0.05)
    lines = out0.read().decode(errors="replace").splitlines()
    if failed is not None:
        sys.stderr.write("bench.py: rank %d exited with code %d\n" % (failed, procs[failed].returncode))
        sys.stderr.write("\n".join(lines) + "\n")
        raise SystemExit(1)
    js = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if len(js) != 1:
        raise SystemExit("bench.py: rank 0 printed %d JSON lines" % len(js))
    print(js[0], flush=True)


def load_package():
    import importlib.util
    name = "ugproj_detector_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "ugproj-detector_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def make_frames_parallel(synth, kind, n, rows, cols, first):
    """n distinct frames (global ids first .. first+n-1), generated on a thread pool (numpy releases the GIL)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    out = np.zeros((n, rows, cols), np.uint8)

    def one(i):
        out[i] = synth.make_frames(kind, 1, rows, cols, first=first + i)[0]
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(one, range(n)))
    return out


def rank_main(args):
    import importlib
    import numpy as np
    import torch   # must come before the HIP library: both bind the same libamdhip64.so.7
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        if not (args.share_gpu and args.backend == "gloo"):
            raise SystemExit("rank %d has no GPU of its own (%d visible): one process per GPU "
                             "(rehearse with --share-gpu --backend gloo)" % (local_rank, ndev))
    devno = local_rank % ndev
    torch.cuda.set_device(devno)
    dev = torch.device("cuda", devno)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = load_package()
    synth = importlib.import_module("ugproj_detector_amd.synth")
    shard = importlib.import_module("ugproj_detector_amd.shard")
    import oracle_lib

    cfg = CONFIGS[args.config]
    rows, cols = cfg["rows"], cfg["cols"]
    names = tuple(args.cascade.split(",")) if args.cascade else cfg["cascades"]
    B = args.batch or cfg["batch"]
    cpaths = [oracle_lib.cascade_path(n) for n in names]
    cascs = [pkg.Cascade(p) for p in cpaths]

    # ---- synthetic frames: rank r owns global frames [r*B, (r+1)*B) of every step's global batch; its
    # `ndist` distinct frames are the global synthetic frames r*ndist .. (r+1)*ndist-1 (SplitMix64 per index)
    ndist = min(args.distinct or B, B)
    distinct = make_frames_parallel(synth, args.kind, ndist, rows, cols, first=rank * ndist)
    frames_np = distinct[np.arange(B) % ndist]
    d_frames = torch.from_numpy(frames_np).to(dev)
    torch.cuda.synchronize()

    # Two pipeline slots (workspaces): batch i+1 is enqueued before batch i is waited for, so the host-side
    # candidate decode / rectangle grouping / record packing of batch i overlaps the kernels of batch i+1.
    # Every step still does all of its work.  A slot holds one context per cascade of the config.
    if args.streams == 1:
        # one explicit (non-default) stream shared by all contexts: kernels of consecutive batches run back
        # to back, never concurrently, so the per-kernel HIP-event times are clean
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
        stream = torch.cuda.current_stream().cuda_stream
    else:
        stream = None   # each context owns a stream: kernels of batch i+1 may overlap the tail of batch i
    slots = []
    for _ in range(2):
        cs = [pkg.Context(casc, device=devno, stream=stream) for casc in cascs]
        if len(cs) == 1:
            cs[0].configure(rows, cols, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], CALL["max_size"],
                            max_batch=B, cand_capacity=4096)
        else:
            # several cascades over the same frames: ONE pyramid + integral pass over the union of their scale ladders
            # (cs[0] leads), then each cascade's own variance + cascade kernels
            pkg.configure_shared(cs, rows, cols, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], CALL["max_size"],
                                 max_batch=B, cand_capacity=4096)
        slots.append(cs)
    alg_bytes = sum(c.algorithmic_bytes() for c in slots[0])      # multi-cascade configs sum the per-cascade figures

    # the gather of box records runs on its own stream: it must not queue behind the kernels of the next
    # batch that are already enqueued on the compute stream
    comm_stream = torch.cuda.Stream(device=dev) if (use_dist and args.backend == "nccl") else None
    gatherer = shard.BoxGatherer(dev if (use_dist and args.backend == "nccl") else None)

    def enqueue(slot):
        for c in slot:
            c.enqueue_device(d_frames.data_ptr(), B)

    def wait_boxes(slot):
        """per frame: the rectangles of every cascade of the config, cascade after cascade"""
        per = [c.wait(B) for c in slot]
        if len(per) == 1:
            return per[0]
        return [np.concatenate([p[f] for p in per]) for f in range(B)]

    def start_gather(boxes, global_step):
        lo = (global_step * world + rank) * B
        if comm_stream is not None:
            with torch.cuda.stream(comm_stream):
                gatherer.start(range(lo, lo + B), boxes)
        else:
            gatherer.start(range(lo, lo + B), boxes)

    def run_steps(nsteps, on_done=None):
        """nsteps batches, software-pipelined two deep; the gather of step i finishes while step i+1 computes.
        Returns (boxes of this rank, gathered dict) of the last step."""
        res = None
        enqueue(slots[0])
        pending_gather = False
        for i in range(nsteps):
            if i + 1 < nsteps:
                enqueue(slots[(i + 1) & 1])
            boxes = wait_boxes(slots[i & 1])
            if pending_gather:
                gatherer.finish()
            start_gather(boxes, i)
            pending_gather = True
            if on_done:
                on_done(slots[i & 1])
            res = boxes
        allb = gatherer.finish()
        return res, allb

    run_steps(args.warmup)

    # ---- parity of what is being timed, before any timing: rank 0 checks frames OF EVERY RANK, as they
    # arrive through the gather, against the CPU oracle (the frames are a pure function of their global index)
    boxes, gathered = run_steps(1)
    parity = None
    if rank == 0:
        ocs = [oracle_lib.OracleCascade(p) for p in cpaths]
        ok, nb, nchk = True, 0, 0
        per_rank = min(cfg["parity_frames"], 2 if world > 1 else 4, ndist)
        for r in range(world):
            for f in range(min(per_rank, ndist)):
                fr = frames_np[f] if r == 0 else synth.make_frames(args.kind, 1, rows, cols, first=r * ndist + f)[0]
                want = np.concatenate([oc.detect(fr, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"])[0]
                                       for oc in ocs])
                got = gathered[r * B + f]          # global frame id of step 0: (0 * world + r) * B + f
                ok = ok and np.array_equal(got, want)
                nb += len(want)
                nchk += 1
        ok = ok and sorted(gathered) == list(range(world * B))
        parity = {"frames_checked": nchk, "ranks_checked": world, "boxes": nb, "bit_exact": bool(ok),
                  "gathered_frames": len(gathered)}
        if not ok and not os.environ.get("HC_DEBUG_MAX_STAGES"):
            raise SystemExit("bench: HIP boxes differ from the oracle -- refusing to report a number")

    timing = not args.no_timing_events
    for s in slots:
        for c in s:
            c.enable_timing(timing)
    kernel_ms = {}

    def collect(slot):
        if timing:
            for c in slot:
                for k, v in c.kernel_ms().items():
                    kernel_ms[k] = kernel_ms.get(k, 0.0) + v

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, collect)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for s in slots:
        for c in s:
            c.enable_timing(False)

    if rank == 0:
        fps = world * B * args.steps / elapsed
        cnames = " + ".join("haarcascade_frontalface_%s" % n for n in names)
        out = {
            "metric": "%dp frames/sec, %s, boxes bit-exact vs CPU oracle" % (rows, cnames),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 integrals, f32 features, f64 stage sums", "data": "synthetic",
            "config": {"workload": "%s, scaleFactor 1.1, minNeighbors 2, minSize 30x30, S3 '%s' frames"
                                   % (cfg["label"] if not args.cascade else "%dx%d gray, %s" % (cols, rows, cnames), args.kind),
                       "frames_per_gpu_per_step": B, "distinct_frames_per_gpu": ndist,
                       "levels": [sum(1 for L in c.levels() if L.nx > 0) for c in slots[0]],
                       "shared_prep_levels": len(slots[0][0].levels()) if len(slots[0]) > 1 else None,
                       "parallelism": "frame-shard x%d, %s all-gather of box records"
                                      % (world, "RCCL" if args.backend == "nccl" else "gloo")},
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
            if tf and args.kind == "planted" and args.config == 2 and not args.cascade:
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
        ns = cfg["cpu_sample"] if args.cpu_sample < 0 else args.cpu_sample
        if ns > 0 and world == 1:
            cores = min(os.cpu_count() or 1, 16)   # the GPU box grants a 16-core CPU share per GPU
            sample = distinct[np.arange(ns) % ndist]
            ocs = [oracle_lib.OracleCascade(p) for p in cpaths]
            t1 = time.perf_counter()
            for oc in ocs:
                oc.detect_many(sample, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], (0, 0), nthreads=cores)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(ns / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": "%d of the same %dx%d '%s' frames through %s, frame-parallel over %d threads, "
                                             "%.1f s wall" % (ns, cols, rows, args.kind, cnames, cores, dt)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)          # before anything that could touch the GPU (torch is not even imported here)
        return
    rank_main(args)


if __name__ == "__main__":
    main()
