"""Worker for tests/test_distributed.py: one rank of a world_size-N gloo job that runs the
frame-shard + box-gather plumbing of bench.py (ugproj-detector_amd/shard.py) on CPU."""
import os
import json
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: E402

conftest.load_package()
import importlib  # noqa: E402

shard = importlib.import_module("ugproj_detector_amd.shard")


def fake_boxes(fid, overflow=True):
    """Deterministic stand-in for a frame's detections (a pure function of the frame id)."""
    n = (fid * 7) % 5
    if overflow and fid % 11 == 0:
        n = 70 + fid                                # some frames overflow the fixed record capacity, by different amounts
    return np.array([[fid, k, 30 + k, 30 + k] for k in range(n)], np.int32).reshape(-1, 4)


def main():
    rank, world, nglobal, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    overflow = len(sys.argv) < 6 or sys.argv[5] != "nooverflow"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = (nglobal + world - 1) // world
    lo, hi = shard.shard_range(nglobal, rank, world)
    ids = list(range(lo, hi)) + [-1] * (per - (hi - lo))         # pad so every rank sends `per` records
    boxes = [fake_boxes(i, overflow) if i >= 0 else np.zeros((0, 4), np.int32) for i in ids]
    g = shard.BoxGatherer()
    got = None
    for _ in range(2):                                            # twice through the same persistent buffers
        g.start(ids, boxes)
        got = g.finish()
    assert (got is not None) == (rank == 0)                       # only the consumer unpacks; the others just took part
    if rank == 0:
        got = {fid: b for fid, b in got.items() if fid != -1}     # padding records
        shard.save_boxes(out, got)                                # SURVEY 8f-4 file format: every box of every frame
        np.save(out + ".rec.npy", g.last_records)
        with open(out + ".json", "w") as f:
            json.dump({"overflow_boxes": g.last_overflow_boxes}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
