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
    for i, (fid, b) in enumerate(zip(frame_ids, boxes_per_frame)):
        b = np.asarray(b, np.int32).reshape(-1, 4)
        rec[i, 0] = fid
        rec[i, 1] = len(b)
        k = min(len(b), MAX_BOXES)
        rec[i, 2:2 + 4 * k] = b[:k].ravel()
    return rec


def pack_overflow(boxes_per_frame):
    """-> int32 array [m, 4]: boxes[K:] of every frame with more than K boxes, in record order."""
    parts = [np.asarray(b, np.int32).reshape(-1, 4)[MAX_BOXES:] for b in boxes_per_frame if len(b) > MAX_BOXES]
    return np.concatenate(parts) if parts else np.zeros((0, 4), np.int32)


def overflow_counts(rec, world_size):
    """Boxes beyond K per rank, from the gathered records (rank order, equal record counts per rank)."""
    rec = np.asarray(rec).reshape(world_size, -1, RECORD_INTS)
    return np.maximum(rec[:, :, 1].astype(np.int64) - MAX_BOXES, 0).sum(axis=1)


def unpack_records(rec, overflow=None):
    """-> dict frame_id -> (count, boxes[n,4]).
    Without `overflow`, n = min(count, K) (the fixed part only).  With `overflow` -- a list of one
    [m_r, 4] array per rank, as returned by gather_boxes -- every frame carries all `count` boxes."""
    rec = np.asarray(rec).reshape(-1, RECORD_INTS)
    out = {}
    if overflow is None:
        for r in rec:
            cnt = int(r[1])
            k = min(cnt, MAX_BOXES)
            out[int(r[0])] = (cnt, r[2:2 + 4 * k].reshape(k, 4).copy())
        return out
    world = len(overflow)
    per = len(rec) // world
    for rank in range(world):
        at = 0
        ov = np.asarray(overflow[rank], np.int32).reshape(-1, 4)
        for r in rec[rank * per:(rank + 1) * per]:
            cnt = int(r[1])
            k = min(cnt, MAX_BOXES)
            boxes = r[2:2 + 4 * k].reshape(k, 4)
            if cnt > MAX_BOXES:
                extra = cnt - MAX_BOXES
                if at + extra > len(ov):
                    raise ValueError("overflow block of rank %d is shorter than its records say" % rank)
                boxes = np.concatenate([boxes, ov[at:at + extra]])
                at += extra
            out[int(r[0])] = (cnt, boxes.copy())
        if at != len(ov):
            raise ValueError("overflow block of rank %d is longer than its records say" % rank)
    return out


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
    return g.finish()


class BoxGatherer:
    """Pipelined form of gather_boxes for a caller that overlaps the exchange with its next batch:
    start() issues the collective on the current stream (device tensors for RCCL: the caller selects
    a communication stream with torch.cuda.stream) and returns at once; finish() waits for it, runs
    the rare overflow round and returns the complete dict.  Staging buffers are persistent and
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
            # with the next batch already enqueued can be a whole batch later (ugproj-detector_amd/csrc/hc_api.cpp, ev_copied).
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

    def finish(self):
        import torch
        world, boxes, _n = self._pending
        self._pending = None
        dist = _dist()
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
            allrec = self._h_out.numpy().copy()
        over = overflow_counts(allrec, world)
        self.last_records = allrec
        self.last_overflow_boxes = int(over.sum())
        if over.sum() == 0:
            return {fid: b for fid, (cnt, b) in unpack_records(allrec, [np.zeros((0, 4), np.int32)] * world).items()}
        mine = pack_overflow(boxes)
        if dist is None:
            blocks = [mine]
        else:
            m = int(over.max())
            buf = torch.zeros((m, 4), dtype=torch.int32)
            buf[:len(mine)] = torch.from_numpy(mine)
            if self.device is not None:
                buf = buf.to(self.device)
            out = torch.empty((world * m, 4), dtype=torch.int32, device=buf.device)
            dist.all_gather_into_tensor(out, buf)
            out = out.cpu().numpy().reshape(world, m, 4)
            blocks = [out[r, :int(over[r])] for r in range(world)]
        return {fid: b for fid, (cnt, b) in unpack_records(allrec, blocks).items()}


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
