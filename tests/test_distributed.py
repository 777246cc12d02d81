"""Multi-process path on CPU: world_size 2 (and 3) over gloo.  Frames shard in contiguous blocks,
no data-path collective; the only exchange is the all-gather of fixed-capacity box records
(ugproj-detector_amd/shard.py), which must return every frame's boxes tagged with its frame id --
the tracker consumes frames in order (reference src/tracker/face_tracker.cc:123-148)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_covers_everything(pkg):
    import importlib
    shard = importlib.import_module("ugproj_detector_amd.shard")
    for n in (0, 1, 7, 64, 65, 1000):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                lo, hi = shard.shard_range(n, r, w)
                got += list(range(lo, hi))
            assert got == list(range(n))


def test_pack_unpack_roundtrip(pkg):
    import importlib
    shard = importlib.import_module("ugproj_detector_amd.shard")
    boxes = [np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.int32), np.zeros((0, 4), np.int32),
             np.arange(4 * 70, dtype=np.int32).reshape(70, 4)]
    rec = shard.pack_records([10, 11, 12], boxes)
    assert rec.shape == (3, shard.RECORD_INTS)
    out = shard.unpack_records(rec)
    assert out[10][0] == 2 and out[10][1].tolist() == boxes[0].tolist()
    assert out[11][0] == 0 and len(out[11][1]) == 0
    assert out[12][0] == 70 and out[12][1].tolist() == boxes[2][:shard.MAX_BOXES].tolist()   # overflow flagged by count


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_gather_of_box_records(pkg, tmp_path, world):
    import importlib
    shard = importlib.import_module("ugproj_detector_amd.shard")
    sys.path.insert(0, HERE)
    import _dist_worker
    nglobal = 37
    out = str(tmp_path / "gathered.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), str(world), str(nglobal), out],
                              env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = shard.unpack_records(np.load(out))
    got.pop(-1, None)                                   # padding records
    assert sorted(got) == list(range(nglobal))          # every frame exactly once, addressable by id
    for fid in range(nglobal):
        want = _dist_worker.fake_boxes(fid)
        cnt, boxes = got[fid]
        assert cnt == len(want)
        assert boxes.tolist() == want[:shard.MAX_BOXES].tolist()


def test_box_file_roundtrip(pkg, tmp_path):
    """frame id -> rect list on disk, in frame order, any number of boxes per frame."""
    import importlib
    shard = importlib.import_module("ugproj_detector_amd.shard")
    boxes = {7: np.array([[1, 2, 30, 30]], np.int32), 3: np.zeros((0, 4), np.int32),
             11: np.arange(4 * 100, dtype=np.int32).reshape(100, 4)}
    p = str(tmp_path / "boxes.hcbx")
    shard.save_boxes(p, boxes)
    got = shard.load_boxes(p)
    assert list(got) == [3, 7, 11]
    for k in boxes:
        assert got[k].tolist() == boxes[k].tolist()
    (tmp_path / "bad").write_bytes(b"nope")
    with pytest.raises(ValueError):
        shard.load_boxes(str(tmp_path / "bad"))
