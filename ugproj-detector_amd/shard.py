"""Frame sharding across GPUs and the gather of detection boxes.

Detection of frame i depends on frame i only (reference src/tracker/face_tracker.cc:290-304
passes only `curr_frame`), so frames shard embarrassingly: one process per GPU, rank r takes a
contiguous block of every global batch, no data-path collective.  The only exchange step is
the gather of per-frame box lists back to rank 0 -- the tracker consumes every box of every
frame in frame order (face_tracker.cc:123-148, :294-302), so records carry their global frame
id and the gather is LOSSLESS.  torch.distributed is the transport: backend "nccl" is RCCL over
xGMI on ROCm, "gloo" on CPU for tests.

Record layout (int32): [frame_id, count, x0, y0, w0, h0, ... x(K-1), ...] with K = MAX_BOXES.
A frame with count > K keeps its first K boxes in the record; boxes K.. travel in a second,
variable-length all-gather whose size every rank derives from the counts of the first one
(no extra size exchange): rank r's overflow block is the concatenation, in record order, of
boxes[K:] of its overflowing frames, padded to the largest block of any rank.  The second
gather only happens for a batch in which some frame overflowed.
"""
from collections.abc import Mapping

import numpy as np

MAX_BOXES = 64
RECORD_INTS = 2 + 4 * MAX_BOXES


def shard_range(nframes_global, rank, world_size):
    """Contiguous block [lo, hi) of global frame ids owned by `rank`."""
    per = (nframes_global + world_size - 1) // world_size
    lo = min(rank * per, nframes_global)
    return lo, min(lo + per, nframes_global)


def pack_records(frame_ids, boxes_per_frame):
    """-> int32 array [n, RECORD_INTS]: the fixed part (first K boxes) of every frame."""
    n = len(frame_ids)
    rec = np.zeros((n, RECORD_INTS), np.int32)
    rec[:, 0] = np.fromiter(frame_ids, np.int32, n)
    cnt = np.fromiter((len(b) for b in boxes_per_frame), np.int32, n)
    rec[:, 1] = cnt
    k = np.minimum(cnt, MAX_BOXES)
    if n and int(k.max()) > 0:
        # all boxes of the batch as one array, then one masked scatter into the records (no per-frame row writes)
        parts = [np.asarray(b, np.int32).reshape(-1, 4)[:MAX_BOXES] for b in boxes_per_frame]
        flat = np.concatenate(parts)
        kmax = int(k.max())
        rec[:, 2:2 + 4 * kmax].reshape(n, kmax, 4)[np.arange(kmax)[None, :] < k[:, None]] = flat
    return rec


def pack_overflow(boxes_per_frame):
    """-> int32 array [m, 4]: boxes[K:] of every frame with more than K boxes, in record order."""
    parts = [np.asarray(b, np.int32).reshape(-1, 4)[MAX_BOXES:] for b in boxes_per_frame if len(b) > MAX_BOXES]
    return np.concatenate(parts) if parts else np.zeros((0, 4), np.int32)


def overflow_counts(rec, world_size):
    """Boxes beyond K per rank, from the gathered records (rank order, equal record counts per rank)."""
    rec = np.asarray(rec).reshape(world_size, -1, RECORD_INTS)
    return np.maximum(rec[:, :, 1].astype(np.int64) - MAX_BOXES, 0).sum(axis=1)


class GatheredBoxes(Mapping):
    """frame id -> boxes[count, 4] of one gathered batch, held COMPACT: one flat array of the boxes the fixed records carry,
    an offset per record, and the overflow pieces of the few frames that have them.  Built with a handful of array
    operations -- no per-record Python -- because rank 0 builds it once per step on the thread that feeds its GPU
    (round 3 unpacked every record of every rank in a Python loop, on every rank: 6 ms per step at world 8).  Nothing is
    materialised per frame until the consumer asks for it; an id that occurs more than once (rank padding records)
    resolves to its first record."""

    def __init__(self, rec, overflow=None):
        rec = np.asarray(rec).reshape(-1, RECORD_INTS)
        n = len(rec)
        self.ids = rec[:, 0].copy()
        self.counts = rec[:, 1].copy()
        if n and int(self.counts.min()) < 0:
            raise ValueError("negative box count in a gathered record")
        k = np.minimum(self.counts, MAX_BOXES)
        self._ofs = np.zeros(n + 1, np.int64)
        np.cumsum(k, out=self._ofs[1:])
        kmax = int(k.max()) if n else 0
        body = rec[:, 2:2 + 4 * kmax].reshape(n, kmax, 4)            # only the columns some record uses
        self._flat = body[np.arange(kmax)[None, :] < k[:, None]]      # [sum k, 4], record order; a copy: the gather buffer is reused
        self._extra = {}
        if overflow is not None:
            world = len(overflow)
            per = n // world if world else 0
            over = np.flatnonzero(self.counts > MAX_BOXES)            # rare: a Python loop over these few is fine
            used = [0] * world
            for i in over.tolist():
                r = min(i // per, world - 1) if per else 0
                ov = np.asarray(overflow[r], np.int32).reshape(-1, 4)
                m = int(self.counts[i]) - MAX_BOXES
                if used[r] + m > len(ov):
                    raise ValueError("overflow block of rank %d is shorter than its records say" % r)
                self._extra[i] = ov[used[r]:used[r] + m].copy()
                used[r] += m
            for r in range(world):
                if used[r] != len(np.asarray(overflow[r]).reshape(-1, 4)):
                    raise ValueError("overflow block of rank %d is longer than its records say" % r)
        self._index = None

    def _lookup(self):
        if self._index is None:
            idx = {}
            for i, f in enumerate(self.ids.tolist()):
                idx.setdefault(f, i)
            self._index = idx
        return self._index

    def record_boxes(self, i):
        b = self._flat[self._ofs[i]:self._ofs[i + 1]]
        e = self._extra.get(i)
        return b if e is None else np.concatenate([b, e])

    def __getitem__(self, fid):
        return self.record_boxes(self._lookup()[fid])

    def __iter__(self):
        return iter(self._lookup())

    def __len__(self):
        return len(self._lookup())

    def count(self, fid):
        return int(self.counts[self._lookup()[fid]])


def unpack_records(rec, overflow=None):
    """-> dict frame_id -> (count, boxes[n,4]).
    Without `overflow`, n = min(count, K) (the fixed part only).  With `overflow` -- a list of one
    [m_r, 4] array per rank, as returned by the second all-gather -- every frame carries all `count` boxes."""
    g = GatheredBoxes(rec, overflow)
    return {fid: (g.count(fid), g[fid].copy()) for fid in g}


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def gather_records(rec, device=None):
    """All-gather of the fixed-size record blocks; every rank must pass the same number of records.
    Returns the concatenated records of all ranks (rank order) as a numpy array on every rank.
    Without an initialised process group it is the identity.  Fixed part only: use gather_boxes
    for the complete, lossless exchange."""
    import torch
    dist = _dist()
    if dist is None:
        return np.asarray(rec)
    t = torch.from_numpy(np.ascontiguousarray(rec))
    if device is not None:
        t = t.to(device, non_blocking=True)
    out = torch.empty((dist.get_world_size() * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


def gather_boxes(frame_ids, boxes_per_frame, device=None):
    """The complete exchange step: -> dict frame_id -> boxes[count, 4] with EVERY box of every frame
    of every rank, on every rank (rank 0 is the consumer).  One all-gather of fixed records, plus a
    second one of the overflow blocks only when some frame of the batch has more than K boxes."""
    g = BoxGatherer(device)
    g.start(frame_ids, boxes_per_frame)
    return g.finish(consume=True)


class BoxGatherer:
    """Pipelined form of gather_boxes for a caller that overlaps the exchange with its next batch:
    start() issues the collective on the current stream (device tensors for RCCL: the caller selects
    a communication stream with torch.cuda.stream) and returns at once; finish() waits for it, runs
    the rare overflow round and returns the complete mapping on rank 0.  Staging buffers are persistent and
    page-locked, so neither direction blocks the host thread that feeds the GPU."""

    def __init__(self, device=None):
        self.device = device
        self._h_in = self._h_out = self._d_in = self._d_out = None
        self._event = None
        self._pending = None
        self._seq = 0
        self._flag_h = self._flag_d = self._flag_np = None

    def _buffers(self, n, world):
        import torch
        if self._h_in is not None and self._h_in.shape[0] == n and self._h_out.shape[0] == n * world:
            return
        pin = self.device is not None
        self._h_in = torch.empty((n, RECORD_INTS), dtype=torch.int32, pin_memory=pin)
        self._h_out = torch.empty((n * world, RECORD_INTS), dtype=torch.int32, pin_memory=pin)
        if self.device is not None:
            self._d_in = torch.empty((n, RECORD_INTS), dtype=torch.int32, device=self.device)
            self._d_out = torch.empty((n * world, RECORD_INTS), dtype=torch.int32, device=self.device)
            self._event = torch.cuda.Event()
            # completion flag: a sequence number copied to page-locked host memory behind the records, polled by finish().
            # Waiting on the event instead returns when the hardware queue the stream was mapped to has drained, which
            # with the next batch already enqueued can be a whole batch later (ugproj-detector_amd/csrc/hc_context.h, ev_copied).
            self._flag_h = torch.zeros(1, dtype=torch.int32, pin_memory=True)
            self._flag_d = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._flag_np = self._flag_h.numpy()

    def start(self, frame_ids, boxes_per_frame):
        import torch
        assert self._pending is None, "previous gather not finished"
        rec = pack_records(frame_ids, boxes_per_frame)
        dist = _dist()
        world = dist.get_world_size() if dist is not None else 1
        self._pending = (world, [np.asarray(b, np.int32).reshape(-1, 4) for b in boxes_per_frame], len(rec))
        if dist is None:
            self._rec_local = rec
            return
        self._buffers(len(rec), world)
        self._h_in.copy_(torch.from_numpy(rec))
        if self.device is not None:
            self._d_in.copy_(self._h_in, non_blocking=True)
            dist.all_gather_into_tensor(self._d_out, self._d_in)
            self._h_out.copy_(self._d_out, non_blocking=True)
            self._seq += 1
            self._flag_d.fill_(self._seq)
            self._flag_h.copy_(self._flag_d, non_blocking=True)
            self._event.record()
        else:
            dist.all_gather_into_tensor(self._h_out, self._h_in)

    def finish(self, consume=None):
        """Waits for the exchange and returns the gathered batch as a GatheredBoxes mapping (frame id -> boxes) on the
        CONSUMER, None elsewhere.  consume=None: rank 0 consumes (the tracker's side, reference
        src/tracker/face_tracker.cc:123-148); the other ranks only read the gathered counts, which they need to size
        the overflow round.  consume=True: this rank wants the mapping too (gather_boxes)."""
        import torch
        world, boxes, _n = self._pending
        self._pending = None
        dist = _dist()
        if consume is None:
            consume = dist is None or dist.get_rank() == 0
        if dist is None:
            allrec = self._rec_local
        else:
            if self.device is not None:
                import time
                deadline = time.monotonic() + 30.0
                while int(self._flag_np[0]) != self._seq:
                    if time.monotonic() > deadline:      # never seen; the event wait then also surfaces a failed collective
                        self._event.synchronize()
                        break
                    time.sleep(5e-5)
            allrec = self._h_out.numpy()                 # the persistent buffer itself: GatheredBoxes copies what it keeps
        return self._complete(allrec, world, boxes, consume)

    def _complete(self, allrec, world, boxes, consume):
        """what finish() does on the host once the records are in `allrec`: the gathered counts (every rank), the overflow
        round if some frame needs it (every rank), the compact mapping (consumer only)"""
        import torch
        dist = _dist()
        over = overflow_counts(allrec, world)
        self.last_records = allrec                       # valid until the next start()
        self.last_overflow_boxes = int(over.sum())
        if over.sum() == 0:
            return GatheredBoxes(allrec) if consume else None
        mine = pack_overflow(boxes)
        if dist is None:
            blocks = [mine] if world == 1 else None
            if blocks is None:
                raise RuntimeError("overflow round of a multi-rank batch needs the process group")
        else:
            m = int(over.max())
            buf = torch.zeros((m, 4), dtype=torch.int32)
            buf[:len(mine)] = torch.from_numpy(mine)
            if self.device is not None:
                buf = buf.to(self.device)
            out = torch.empty((world * m, 4), dtype=torch.int32, device=buf.device)
            dist.all_gather_into_tensor(out, buf)         # every rank takes part, whoever consumes
            if not consume:
                return None
            out = out.cpu().numpy().reshape(world, m, 4)
            blocks = [out[r, :int(over[r])] for r in range(world)]
        return GatheredBoxes(allrec, blocks) if consume else None


# ---- on-disk form of the same records, for handing detections to the sequential tracker /
# clusterer out of band (the reference only has a TODO for tracker metadata files, src/file_io.h:41-48).
# Layout (little endian): magic "HCBX", int32 version = 1, int32 nframes, then per frame
# int32 frame_id, int32 count, count x int32[4] (x, y, width, height).  Frames are stored in ascending id.
# The C ABI reads and writes the same bytes (hc_boxes_save / hc_boxes_load, include/hip_cascade.h).
_MAGIC = b"HCBX"


def save_boxes(path, boxes_by_frame):
    """boxes_by_frame: dict frame_id -> (K,4) int array (all K boxes, no capacity limit)"""
    import struct
    with open(path, "wb") as f:
        f.write(_MAGIC + struct.pack("<ii", 1, len(boxes_by_frame)))
        for fid in sorted(boxes_by_frame):
            b = np.asarray(boxes_by_frame[fid], np.int32).reshape(-1, 4)
            f.write(struct.pack("<ii", int(fid), len(b)))
            f.write(b.astype("<i4").tobytes())


def load_boxes(path):
    import struct
    data = open(path, "rb").read()
    if data[:4] != _MAGIC:
        raise ValueError("not an HCBX file")
    ver, n = struct.unpack_from("<ii", data, 4)
    if ver != 1:
        raise ValueError("unsupported HCBX version %d" % ver)
    p, out = 12, {}
    for _ in range(n):
        fid, cnt = struct.unpack_from("<ii", data, p)
        p += 8
        out[fid] = np.frombuffer(data, "<i4", cnt * 4, p).reshape(cnt, 4).copy()
        p += cnt * 16
    return out
