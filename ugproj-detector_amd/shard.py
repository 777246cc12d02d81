"""Frame sharding across GPUs and the gather of detection boxes.

Detection of frame i depends on frame i only (reference src/tracker/face_tracker.cc:290-304
passes only `curr_frame`), so frames shard embarrassingly: one process per GPU, rank r takes a
contiguous block of every global batch, no data-path collective.  The only exchange step is
the gather of per-frame box lists back to rank 0 -- the tracker consumes frames in order
(face_tracker.cc:123-148), so records carry their global frame id.  torch.distributed is the
transport: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests.

Record layout (int32): [frame_id, count, x0, y0, w0, h0, ... x(K-1), ...] with K = MAX_BOXES.
count > K flags an overflow; the boxes beyond K are then fetched point-to-point.
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
    """-> int32 array [n, RECORD_INTS]"""
    n = len(frame_ids)
    rec = np.zeros((n, RECORD_INTS), np.int32)
    for i, (fid, b) in enumerate(zip(frame_ids, boxes_per_frame)):
        b = np.asarray(b, np.int32).reshape(-1, 4)
        rec[i, 0] = fid
        rec[i, 1] = len(b)
        k = min(len(b), MAX_BOXES)
        rec[i, 2:2 + 4 * k] = b[:k].ravel()
    return rec


def unpack_records(rec):
    """-> dict frame_id -> (count, boxes[min(count,K),4])"""
    out = {}
    for r in np.asarray(rec).reshape(-1, RECORD_INTS):
        cnt = int(r[1])
        k = min(cnt, MAX_BOXES)
        out[int(r[0])] = (cnt, r[2:2 + 4 * k].reshape(k, 4).copy())
    return out


def gather_records(rec, device=None):
    """All-gather fixed-size record blocks; every rank must pass the same number of records.
    Returns the concatenated records of all ranks (rank order) as a numpy array on every rank
    (rank 0 is the consumer).  Without an initialised process group it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(rec)
    t = torch.from_numpy(np.ascontiguousarray(rec))
    if device is not None:
        t = t.to(device, non_blocking=True)
    out = torch.empty((dist.get_world_size() * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy()


# ---- on-disk form of the same records, for handing detections to the sequential tracker /
# clusterer out of band (the reference only has a TODO for tracker metadata files, src/file_io.h:41-48).
# Layout (little endian): magic "HCBX", int32 version = 1, int32 nframes, then per frame
# int32 frame_id, int32 count, count x int32[4] (x, y, width, height).  Frames are stored in ascending id.
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
