#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: 720p frames/sec, haarcascade_frontalface_alt, boxes bit-exact.

A "step" = one pass of the hot path (pyramid -> integrals -> variance -> cascade -> box
grouping) over one batch of synthetic gray frames per GPU, frames already resident in HBM when
the timed region starts.  Workloads (`--config`, BASELINE.json `configs`):
  2 (default, the configuration the metric is quoted on): 1280x720, alt, scaleFactor 1.1,
    minNeighbors 2, minSize 30x30 -- the literal call at reference src/tracker/detector.cc:25-31;
  3: 1920x1080, alt2 + default on the same frames (two-cascade pass);
  5: 3840x2160, all four data/ cascades on the same frames.
The default single-GPU run measures config 2 (that is `value`) and then, outside its timed
region, a few steps of configs 3 and 5 with their own oracle check (`extra.configs`).
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
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))      # oracle_lib: the checker and the cpu_baseline leg only

CALL = dict(scale_factor=1.1, min_neighbors=2, min_size=(30, 30), max_size=(0, 0))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    2: dict(rows=720, cols=1280, cascades=("alt",), batch=512, cpu_sample=256, parity_frames=16,
            label="BASELINE config #2: 1280x720 gray, haarcascade_frontalface_alt"),
    3: dict(rows=1080, cols=1920, cascades=("alt2", "default"), batch=384, cpu_sample=32, parity_frames=4,
            label="BASELINE config #3: 1920x1080 gray, haarcascade_frontalface_alt2 + _default on the same frames"),
    5: dict(rows=2160, cols=3840, cascades=("alt", "alt2", "alt_tree", "default"), batch=128, cpu_sample=8, parity_frames=2,
            label="BASELINE config #5: 3840x2160 gray, all four data/ cascades on the same frames"),
}
EXTRA_CONFIGS = (3, 5)         # measured after the headline, a few steps each (`extra.configs`)


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
    ap.add_argument("--semantics", default="cv4", choices=["cv4", "cv24"],
                    help="cv4: OpenCV >= 4 evaluation (BASELINE metric); cv24: the OpenCV 2.4.x evaluation of the same call "
                         "(hc_context_set_semantics, SURVEY Appendix B) -- one cascade per run")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="frames of the CPU-oracle baseline sample (0 = skip)")
    ap.add_argument("--no-timing-events", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra.configs leg (configs 3 and 5 after the headline)")
    ap.add_argument("--extra-steps", type=int, default=5)
    ap.add_argument("--streams", type=int, default=1, choices=[1, 2],
                    help="1: both pipeline slots on one stream (clean per-kernel times); 2: a stream per slot")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="transport of the box gather: nccl = RCCL over xGMI; gloo = rehearsal on a box with fewer GPUs than ranks")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks share the visible GPUs (rank r -> device r mod count); needs --backend gloo")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the gather even with one rank")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="seconds before self-spawned ranks are terminated")
    return ap.parse_args(argv)


def rendezvous_store(world):
    """The rendezvous of self-spawned ranks: a TCPStore SERVER in the parent, on a port the OS picks at bind time and this
    process keeps bound for the whole run -- there is no window between choosing a port and a rank binding it (round 3
    closed a probe socket and hoped the port stayed free).  torch.distributed's store is CPU code: no HIP call here."""
    from datetime import timedelta
    from torch.distributed import TCPStore
    return TCPStore("127.0.0.1", 0, world, is_master=True, timeout=timedelta(seconds=600), wait_for_workers=False)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same script
    (fresh interpreters, this parent never initialises HIP and never execs), relay rank 0's JSON
    line, fail if any rank fails or the run exceeds --spawn-timeout."""
    env = dict(os.environ)
    store = rendezvous_store(args.gpus)          # held until every rank has exited
    env.update(WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(store.port),
               HC_BENCH_STORE_PORT=str(store.port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # rank 0's stdout is drained while the ranks run: a rank that prints more than a pipe buffer (library or RCCL
    # debug output) would otherwise block in write() with the other ranks waiting for it in a collective
    lines = []

    def drain():
        for raw in procs[0].stdout:
            lines.append(raw.decode(errors="replace").rstrip("\n"))
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()
    failed, timed_out = None, False
    deadline = time.monotonic() + args.spawn_timeout
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if failed is None and (bad or time.monotonic() > deadline):
            failed = bad[0] if bad else -1
            timed_out = not bad
            for p in procs:                      # exact PIDs we started, nothing else
                if p.poll() is None:
                    p.terminate()
            deadline = time.monotonic() + 20
        elif failed is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        if all(c is not None for c in codes):
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    del store
    if failed is not None:
        if timed_out:
            sys.stderr.write("bench.py: ranks still running after %.0f s -- terminated\n" % args.spawn_timeout)
        else:
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


def cpu_budget():
    """Host cores this process tree may really use: the cgroup CPU quota (the GPU box grants 16 cores per GPU as a CFS
    quota while os.cpu_count() still reports every core of the host -- 256), the affinity mask, the core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def host_threads(world):
    """this rank's share of the host cores it may use (quota-aware): frame generator, oracle checker"""
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "0")) or max(world, 1)
    return max(1, min(16, cpu_budget() // local))


def tame_thread_pools(world):
    """Before torch / numpy are imported: their OpenMP / BLAS pools size themselves by os.cpu_count() (256 on the GPU
    box) and spin after every parallel region; a few small CPU tensor operations per step then burn the whole CFS quota
    and the kernel throttles every thread of the rank for tens of milliseconds (measured with one rank and RCCL:
    nr_throttled 16, 30.6 instead of 15.1 ms per step).  Nothing on the timed path needs a CPU math pool."""
    n = str(max(1, min(4, host_threads(world))))
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(k, n)
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    os.environ.setdefault("GOMP_SPINCOUNT", "0")


def spread_indices(B, n):
    """n batch positions to check against the oracle, chosen so that EVERY kind of place a frame can have in a launch is
    covered: first and last frame, both sides of every 128-frame boundary (k_cascade's frame chunks) and of 32-frame
    boundaries (k_pyramid_integral's), then evenly spaced fill.  Round 3 checked frames 0..15 -- chunk 0 of both remaps."""
    cand = [0, B - 1]
    for c in (128, 32):
        for k in range(1, (B + c - 1) // c):
            cand += [k * c - 1, k * c]
    step = max(1, B // max(n, 1))
    cand += [(i * step + step // 2) % B for i in range(n)]
    out = []
    for i in cand:
        if 0 <= i < B and i not in out:
            out.append(i)
    # boundaries of the big chunks first, but keep the fill in when the batch has many boundaries
    if len(out) > n:
        head = out[:2] + [i for i in out[2:] if (i % 128) in (0, 127)]
        rest = [i for i in out if i not in head]
        out = (head + rest)[:n] if len(head) <= n else head[:n]
    return sorted(out)


def make_frames_parallel(synth, kind, n, rows, cols, first, nthreads):
    """n distinct frames (global ids first .. first+n-1), generated on a thread pool (numpy releases the GIL)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    out = np.zeros((n, rows, cols), np.uint8)

    def one(i):
        out[i] = synth.make_frames(kind, 1, rows, cols, first=first + i)[0]
    with ThreadPoolExecutor(max_workers=nthreads) as ex:
        list(ex.map(one, range(n)))
    return out


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block: RCCL prints a version banner to stdout when a communicator
    is made, and rank 0's stdout carries exactly one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Env:
    """what one rank process sets up once: device, process group, modules"""
    pass


def setup(args):
    import importlib
    tame_thread_pools(int(os.environ.get("WORLD_SIZE", "1")))
    import torch   # must come before the HIP library: both bind the same libamdhip64.so.7
    import torch.distributed as dist

    E = Env()
    E.torch, E.dist = torch, dist
    E.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    E.world = int(os.environ.get("WORLD_SIZE", "1"))
    if E.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, E.world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    E.ndev = torch.cuda.device_count()
    if local_rank >= E.ndev:
        if not (args.share_gpu and args.backend == "gloo"):
            raise SystemExit("rank %d has no GPU of its own (%d visible): one process per GPU "
                             "(rehearse with --share-gpu --backend gloo)" % (local_rank, E.ndev))
    E.devno = local_rank % E.ndev
    E.shared = E.world > E.ndev                      # several ranks per device: a rehearsal, never a scaling point
    torch.cuda.set_device(E.devno)
    E.dev = torch.device("cuda", E.devno)
    E.use_dist = E.world > 1 or args.force_dist
    E.nthreads = host_threads(E.world)
    torch.set_num_threads(max(1, min(4, E.nthreads)))
    # the library's own post-processing pool (candidate decode + grouping per frame) gets its share too
    os.environ.setdefault("HC_HOST_THREADS", str(max(1, min(4, E.nthreads))))
    if E.use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # self-spawned ranks meet at the store the parent serves (its port was bound before any rank started);
        # ranks started by torch.distributed.run use the launcher's own rendezvous (env://)
        kw = {}
        if os.environ.get("HC_BENCH_STORE_PORT"):
            from datetime import timedelta
            kw["store"] = dist.TCPStore("127.0.0.1", int(os.environ["HC_BENCH_STORE_PORT"]), E.world, is_master=False,
                                        timeout=timedelta(seconds=600))
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=E.rank, world_size=E.world, device_id=E.dev, **kw)
                dist.barrier()             # the communicator (and its banner) now, not inside a measured region
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo", rank=E.rank, world_size=E.world, **kw)
    E.pkg = load_package()
    E.synth = importlib.import_module("ugproj_detector_amd.synth")
    E.shard = importlib.import_module("ugproj_detector_amd.shard")
    import oracle_lib                               # test infrastructure: parity check before timing, cpu_baseline
    E.oracle = oracle_lib
    if args.streams == 1:
        # one explicit (non-default) stream shared by all contexts: kernels of consecutive batches run back
        # to back, never concurrently, so the per-kernel HIP-event times are clean
        torch.cuda.set_stream(torch.cuda.Stream(device=E.dev))
        E.stream = torch.cuda.current_stream().cuda_stream
    else:
        E.stream = None   # each context owns a stream: kernels of batch i+1 may overlap the tail of batch i
    return E


def run_config(E, args, cfgno, steps, warmup, batch=0, names=None, cpu_sample=-1, timing=True, kind=None, distinct=0):
    """One workload, measured: returns the result dict on rank 0 (None elsewhere)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    torch, dist, pkg, synth, shard, oracle_lib = E.torch, E.dist, E.pkg, E.synth, E.shard, E.oracle
    rank, world, dev = E.rank, E.world, E.dev
    cfg = CONFIGS[cfgno]
    rows, cols = cfg["rows"], cfg["cols"]
    names = tuple(names) if names else cfg["cascades"]
    kind = kind or args.kind
    B = batch or cfg["batch"]
    cpaths = [pkg.cascade_path(n) for n in names]
    cascs = [pkg.Cascade(p) for p in cpaths]

    # ---- synthetic frames: rank r owns global frames [r*B, (r+1)*B) of every step's global batch; its
    # `ndist` distinct frames are the global synthetic frames r*ndist .. (r+1)*ndist-1 (SplitMix64 per index)
    ndist = min(distinct or args.distinct or B, B)
    distinct = make_frames_parallel(synth, kind, ndist, rows, cols, first=rank * ndist, nthreads=E.nthreads)
    frames_np = distinct[np.arange(B) % ndist]
    d_frames = torch.from_numpy(frames_np).to(dev)
    torch.cuda.synchronize()

    # Two pipeline slots (workspaces): batch i+1 is enqueued before batch i is waited for, so the host-side
    # candidate decode / rectangle grouping / record packing of batch i overlaps the kernels of batch i+1.
    # Every step still does all of its work.  A slot holds one context per cascade of the config.
    cv24 = args.semantics == "cv24"
    if cv24 and len(cascs) != 1:
        raise SystemExit("--semantics cv24 runs one cascade per context (no shared prep pass): use --cascade")
    slots = []
    for _ in range(2):
        cs = [pkg.Context(casc, device=E.devno, stream=E.stream) for casc in cascs]
        if cv24:
            cs[0].set_semantics(True)
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
    rccl = E.use_dist and args.backend == "nccl"
    comm_stream = torch.cuda.Stream(device=dev) if rccl else None
    gatherer = shard.BoxGatherer(dev if rccl else None)

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
        trace = [] if os.environ.get("HC_BENCH_TRACE") else None      # host-side time per pipeline stage, to stderr
        enqueue(slots[0])
        pending_gather = False
        for i in range(nsteps):
            t = [time.perf_counter()]
            if i + 1 < nsteps:
                enqueue(slots[(i + 1) & 1])
            t.append(time.perf_counter())
            boxes = wait_boxes(slots[i & 1])
            t.append(time.perf_counter())
            if pending_gather:
                gatherer.finish()
            t.append(time.perf_counter())
            start_gather(boxes, i)
            t.append(time.perf_counter())
            pending_gather = True
            if on_done:
                on_done(slots[i & 1])
            res = boxes
            if trace is not None:
                trace.append([1e3 * (b - a) for a, b in zip(t, t[1:])])
        allb = gatherer.finish()
        if trace:
            for i, r in enumerate(trace):
                print("[trace] rank %d step %3d: enqueue %.2f  wait %.2f  gather.finish %.2f  gather.start %.2f ms" % ((rank, i) + tuple(r)),
                      file=sys.stderr)
        return res, allb

    if warmup > 0:
        run_steps(warmup)

    # ---- parity of what is being timed, before any timing: rank 0 checks frames OF EVERY RANK, as they
    # arrive through the gather, against the CPU oracle (the frames are a pure function of their global index)
    boxes, gathered = run_steps(1)
    parity = None
    if rank == 0:
        ocs = [oracle_lib.OracleCascade(p) for p in cpaths]
        per_rank = min(cfg["parity_frames"] if world == 1 else max(2, 16 // world), B)
        # batch positions spread over every frame chunk of both kernels (not the first frames of chunk 0); at N > 1 each
        # rank is checked at ITS OWN positions of the spread, so the ranks together cover the batch
        allpos = spread_indices(B, max(per_rank * world, per_rank))
        jobs = [(r, allpos[(r * per_rank + j) % len(allpos)]) for r in range(world) for j in range(per_rank)]

        def check(job):
            r, f = job
            fr = frames_np[f] if r == 0 else synth.make_frames(kind, 1, rows, cols, first=r * ndist + f % ndist)[0]
            if cv24:
                per = [ocs[0].detect_cv24(fr, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"])[0]]
            elif len(jobs) >= E.nthreads or len(ocs) == 1:
                per = [oc.detect(fr, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"])[0] for oc in ocs]
            else:        # few large frames, several cascades (config 5): the cascades of a frame side by side
                with ThreadPoolExecutor(max_workers=len(ocs)) as ex2:
                    per = list(ex2.map(lambda oc: oc.detect(fr, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"])[0], ocs))
            want = np.concatenate(per)
            got = gathered[r * B + f]          # global frame id of step 0: (0 * world + r) * B + f
            return bool(np.array_equal(got, want)), len(want)
        with ThreadPoolExecutor(max_workers=E.nthreads) as ex:      # the oracle call releases the GIL
            res = list(ex.map(check, jobs))
        ok = all(r[0] for r in res) and sorted(gathered) == list(range(world * B))
        parity = {"frames_checked": len(jobs), "ranks_checked": world, "boxes": int(sum(r[1] for r in res)),
                  "bit_exact": bool(ok), "gathered_frames": len(gathered),
                  "batch_positions": [f for r, f in jobs if r == 0]}
        if not ok and not os.environ.get("HC_DEBUG_MAX_STAGES"):
            raise SystemExit("bench: HIP boxes differ from the oracle -- refusing to report a number")

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
        if E.use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    run_steps(steps, collect)
    own = time.perf_counter() - t0             # this rank's own K steps (before it waits for the others)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_s = [own]
    if E.use_dist:
        cdev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([own], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_s = [float(x.item()) for x in every]
    for s in slots:
        for c in s:
            c.enable_timing(False)

    out = None
    if rank == 0:
        fps = world * B * steps / elapsed
        cnames = " + ".join("haarcascade_frontalface_%s" % n for n in names)
        if not E.use_dist:
            par = "single GPU: no process group, no collective (box records packed and unpacked locally)"
        else:
            par = "frame-shard x%d, %s all-gather of box records" % (world, "RCCL" if args.backend == "nccl" else "gloo")
        out = {
            "metric": "%dp frames/sec, %s%s, boxes bit-exact vs CPU oracle" % (rows, cnames, ", OpenCV 2.4.x semantics" if cv24 else ""),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": min(world, E.ndev) if E.shared else world,
            "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i32 integrals, f32 features, f64 stage sums", "data": "synthetic",
            "config": {"workload": "%s, scaleFactor 1.1, minNeighbors 2, minSize 30x30, S3 '%s' frames"
                                   % (cfg["label"] if names == cfg["cascades"] else "%dx%d gray, %s" % (cols, rows, cnames), kind),
                       "frames_per_gpu_per_step": B, "distinct_frames_per_gpu": ndist,
                       "levels": [sum(1 for L in c.levels() if L.nx > 0) for c in slots[0]],
                       "shared_prep_levels": len(slots[0][0].levels()) if len(slots[0]) > 1 else None,
                       "parallelism": par,
                       # what the library reports it launched for the last timed batch, per cascade of the config
                       # (hc_context_last_launch): kernel build per scan-step class, prep / variance route, frame chunks
                       "launch": [dict(cascade=n, **{k: v for k, v in c.last_launch().items() if k != "nframes"})
                                  for n, c in zip(names, slots[0])]},
            "parity": parity,
            "per_rank_fps": [round(B * steps / s, 1) for s in rank_s],
        }
        if E.shared:
            out["rehearsal"] = True              # ranks share a device: exercises the N > 1 code path, says nothing about scaling
            out["ranks"] = world
        if timing and kernel_ms and not E.shared:
            per_launch = {k: v / steps for k, v in kernel_ms.items()}
            dom = max(per_launch, key=per_launch.get)
            achieved = alg_bytes * B / (per_launch[dom] * 1e-3) / 1e9
            # HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
            # command (tools/prof_round.sh -> tools/traffic_from_pmc.py); PMC cannot be collected from
            # inside this process, so the figure is read from profiles/ and is null when absent/stale.
            traffic, traffic_src = None, None
            import glob
            tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
            if tf and kind == "planted" and cfgno == 2 and names == cfg["cascades"] and not cv24:
                try:
                    tj = json.load(open(tf[-1]))
                    traffic = int(tj["kernels"][dom]["hbm_bytes_per_launch"] * B / tj.get("frames_per_launch", 64))
                    traffic_src = os.path.relpath(tf[-1], ROOT)
                except Exception:
                    traffic = None
            # LDS / VALU counters of the dominant kernel (SURVEY 8d asks for them next to the roofline fraction): from the
            # committed PMC passes of this same command, like `traffic` (tools/prof_round.sh -> tools/pmc_summary.py --json)
            lds = None
            lf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_lds.json")))
            if lf and kind == "planted" and cfgno == 2 and names == cfg["cascades"] and not cv24:
                try:
                    lj = json.load(open(lf[-1]))
                    lds = dict(lj[dom], source=os.path.relpath(lf[-1], ROOT))
                except Exception:
                    lds = None
            out["roofline"] = {"bound": "hbm", "kernel": dom, "lds": lds, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": alg_bytes * B,
                               "launch_ms": round(per_launch[dom], 4),
                               "all_kernels_ms_per_launch": {k: round(v, 4) for k, v in per_launch.items()}}
        ns = cfg["cpu_sample"] if cpu_sample < 0 else cpu_sample
        if cv24:
            ns = 0      # the oracle's 2.4 restatement is single-frame; the cpu_baseline leg times the >= 4 path only
        if ns > 0:
            # reported at any N: rank 0's host cores, the same frames (the other ranks wait at the closing barrier)
            cores = min(cpu_budget(), 16)          # the GPU box grants a 16-core CPU share (CFS quota) per GPU
            sample = distinct[np.arange(ns) % ndist]
            ocs = [oracle_lib.OracleCascade(p) for p in cpaths]
            t1 = time.perf_counter()
            for oc in ocs:
                oc.detect_many(sample, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], (0, 0), nthreads=cores)
            dt = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": round(ns / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": "%d of the same %dx%d '%s' frames through %s, frame-parallel over %d threads, "
                                             "%.1f s wall" % (ns, cols, rows, kind, cnames, cores, dt)}
    for s in slots:
        for c in s:
            c.close()
    del d_frames
    torch.cuda.empty_cache()
    return out


def run_e2e(E, args, batches=12, B=256):
    """end to end -- page-locked host frames -> H2D (0.92 MB per frame) -> detect -> grouped boxes on the
    host, config #2, through hc_detect_batch_enqueue_host with two pipeline slots on one stream (batch i+1 goes up on the
    slot's upload stream while batch i computes).  Never the bench `value`; reported as extra.e2e."""
    import ctypes as C
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    pkg, synth, oracle_lib = E.pkg, E.synth, E.oracle
    cfg = CONFIGS[2]
    rows, cols = cfg["rows"], cfg["cols"]
    path = pkg.cascade_path("alt")
    casc = pkg.Cascade(path)
    frames = make_frames_parallel(synth, "planted", 2 * B, rows, cols, first=0, nthreads=E.nthreads)
    pools = []
    for k in range(2):                      # the caller's frame pool: page-locked (hc_host_alloc), as a decoder would keep it
        ptr = C.c_void_p()
        pkg._check(pkg.lib.hc_host_alloc(B * rows * cols, C.byref(ptr)))
        a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(B, rows, cols))
        a[:] = frames[k * B:(k + 1) * B]
        pools.append((ptr, a))
    ctxs = []
    for k in range(2):
        c = pkg.Context(casc, device=E.devno, stream=ctxs[0].stream() if ctxs else None)
        c.configure(rows, cols, CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"], CALL["max_size"], max_batch=B,
                    cand_capacity=4096)
        ctxs.append(c)

    def run(n):
        res = [None, None]
        ctxs[0].enqueue_host(pools[0][1])
        for i in range(n):
            if i + 1 < n:
                ctxs[(i + 1) & 1].enqueue_host(pools[(i + 1) & 1][1])
            res[i & 1] = ctxs[i & 1].wait(B)
        return res
    run(2)
    E.torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(batches)
    dt = time.perf_counter() - t0
    oc = oracle_lib.OracleCascade(path)
    pos = [(k, f) for k in range(2) for f in spread_indices(B, 4)]
    with ThreadPoolExecutor(max_workers=E.nthreads) as ex:
        want = list(ex.map(lambda kf: oc.detect(frames[kf[0] * B + kf[1]], CALL["scale_factor"], CALL["min_neighbors"], CALL["min_size"])[0], pos))
    ok = all(np.array_equal(res[k][f], w) for (k, f), w in zip(pos, want))
    for c in ctxs:
        c.close()
    for ptr, _ in pools:
        pkg.lib.hc_host_free(ptr)
    if not ok:
        raise SystemExit("bench: e2e boxes differ from the oracle")
    return {"value": round(batches * B / dt, 1), "unit": "frames/s", "batches": batches, "frames_per_batch": B,
            "ms_per_batch": round(1e3 * dt / batches, 3),
            "path": "page-locked host frames -> hc_detect_batch_enqueue_host (H2D on the slot's upload stream) -> kernels -> "
                    "published candidates + boxes grouped on the device (K6); two slots on one stream",
            "parity": {"frames_checked": len(pos), "bit_exact": bool(ok)}}


def rank_main(args):
    E = setup(args)
    names = tuple(args.cascade.split(",")) if args.cascade else None
    out = run_config(E, args, args.config, args.steps, args.warmup, batch=args.batch, names=names, cpu_sample=args.cpu_sample,
                     timing=not args.no_timing_events)
    # Driver-visible numbers for BASELINE configs #3 and #5: only in the plain single-GPU default run, after the headline
    # measurement, never inside its timed region; a few steps each with their own oracle check.
    plain = (E.world == 1 and not args.force_dist and args.config == 2 and not args.cascade and not args.batch and args.semantics == "cv4"
             and not args.distinct and args.kind == "planted" and not args.no_extra
             and not os.environ.get("HC_DEBUG_MAX_STAGES"))
    if plain:
        extra = {}
        for cno in EXTRA_CONFIGS:
            t0 = time.perf_counter()
            try:
                r = run_config(E, args, cno, args.extra_steps, 1, cpu_sample=0, timing=True)
                extra[str(cno)] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"],
                                   "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                                   "frames_per_step": r["config"]["frames_per_gpu_per_step"],
                                   "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "launch_ms",
                                                                              "all_kernels_ms_per_launch")},
                                   "parity": r["parity"], "wall_s": round(time.perf_counter() - t0, 1)}
            except BaseException as e:       # the headline line is still printed; the failure is part of the record
                extra[str(cno)] = {"error": "%s: %s" % (type(e).__name__, e)}
        # the S1 noise / S2 texture families next to the S3 headline (a few steps each, own oracle check,
        # 128 distinct frames cycled through the 512 batch slots), and the end-to-end figure from host frames
        kinds = {}
        for kd in ("noise", "texture"):
            t0 = time.perf_counter()
            try:
                r = run_config(E, args, 2, args.extra_steps, 1, cpu_sample=0, timing=True, kind=kd, distinct=128)
                kinds[kd] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                             "frames_per_step": r["config"]["frames_per_gpu_per_step"], "distinct_frames": 128,
                             "roofline": {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "launch_ms")},
                             "parity": r["parity"], "wall_s": round(time.perf_counter() - t0, 1)}
            except BaseException as e:
                kinds[kd] = {"error": "%s: %s" % (type(e).__name__, e)}
        t0 = time.perf_counter()
        try:
            e2e = run_e2e(E, args)
            e2e["wall_s"] = round(time.perf_counter() - t0, 1)
        except BaseException as e:
            e2e = {"error": "%s: %s" % (type(e).__name__, e)}
        if out is not None:
            out["extra"] = {"configs": extra, "kinds": kinds, "e2e": e2e}
    if E.rank == 0:
        print(json.dumps(out), flush=True)
    if E.use_dist:
        E.dist.barrier()
        E.dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)          # before anything that could touch the GPU (torch is not even imported here)
        return
    rank_main(args)


if __name__ == "__main__":
    main()
