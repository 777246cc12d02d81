"""Deterministic synthetic 8-bit gray frames for tests and bench.py.

All arithmetic is integer (numpy uint64 / Python int), so frames are bit-identical on
every machine.  PRNG: SplitMix64, frame i seeded with 0x9E3779B97F4A7C15 * (i + 1).

  S1 "noise"    iid uniform bytes                (best case for early exit, x-skip everywhere)
  S2 "texture"  4 octaves of box-blurred noise, stretched to mean 128 / sigma ~40
                (natural-image-like stage-depth distribution)
  S3 "planted"  S2 + P face patches per frame stamped at SplitMix-chosen places and sizes
                (non-empty detections; every kernel and the grouping are exercised)

The face patch is data/face64.u8: a 64x64 gray crop of the public-domain NASA
portrait shipped with scikit-image (`astronaut.png`), see data/README.md.  It is an
input stimulus, not an oracle.
"""
import os

import numpy as np

GAMMA = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1
_FACE = None


def splitmix64(seed, n):
    """n successive SplitMix64 outputs for `seed` as uint64."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & MASK) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(GAMMA)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def frame_seed(i, salt=0):
    return (GAMMA * (i + 1) + salt * 0xD1B54A32D192ED03) & MASK


def noise(rows, cols, seed):
    n = rows * cols
    words = splitmix64(seed, (n + 7) // 8)
    return words.view(np.uint8)[:n].reshape(rows, cols).copy()


def _box_mean(img, r):
    """integer box mean with edge replication, window (2r+1)^2"""
    p = np.pad(img.astype(np.int64), r, mode="edge")
    ii = np.zeros((p.shape[0] + 1, p.shape[1] + 1), np.int64)
    ii[1:, 1:] = p.cumsum(0).cumsum(1)
    k = 2 * r + 1
    s = ii[k:, k:] - ii[:-k, k:] - ii[k:, :-k] + ii[:-k, :-k]
    return s // (k * k)


def _isqrt(n):
    import math
    return math.isqrt(n)


def texture(rows, cols, seed):
    t = np.zeros((rows, cols), np.int64)
    for o, r in enumerate((2, 4, 8, 16)):
        t += _box_mean(noise(rows, cols, seed + 0x1000 * (o + 1)), r) * (o + 1)
    n = rows * cols
    s1 = int(t.sum())
    s2 = int((t * t).sum())
    mean = s1 // n
    var_fx = (s2 * n - s1 * s1) * 65536 // (n * n)          # variance in 16.16
    std_fx = max(_isqrt(var_fx), 1)                         # sigma in 8.8
    out = 128 + ((t - mean) * (40 * 256)) // std_fx
    return np.clip(out, 0, 255).astype(np.uint8)


def face_patch():
    global _FACE
    if _FACE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "face64.u8")
        _FACE = np.fromfile(path, np.uint8).reshape(64, 64)
    return _FACE


def _scale_patch(patch, size):
    """integer area/nearest resample of the square patch to size x size"""
    n = patch.shape[0]
    if size >= n:
        idx = (np.arange(size) * n) // size
        return patch[np.ix_(idx, idx)]
    # box average over source cells (integer)
    edges = (np.arange(size + 1) * n) // size
    ii = np.zeros((n + 1, n + 1), np.int64)
    ii[1:, 1:] = patch.astype(np.int64).cumsum(0).cumsum(1)
    y0, y1 = edges[:-1, None], edges[1:, None]
    x0, x1 = edges[None, :-1], edges[None, 1:]
    s = ii[y1, x1] - ii[y0, x1] - ii[y1, x0] + ii[y0, x0]
    area = (y1 - y0) * (x1 - x0)
    return ((s + area // 2) // area).astype(np.uint8)


def planted(rows, cols, seed, nfaces=8, min_size=40, max_size=None):
    img = texture(rows, cols, seed)
    max_size = max_size or max(min(rows, cols) // 3, min_size + 1)
    rnd = splitmix64(seed ^ 0xFACE, 3 * nfaces)
    boxes = []
    face = face_patch()
    for k in range(nfaces):
        size = min_size + int(rnd[3 * k] % np.uint64(max_size - min_size))
        if size > rows or size > cols:
            continue
        x = int(rnd[3 * k + 1] % np.uint64(cols - size + 1))
        y = int(rnd[3 * k + 2] % np.uint64(rows - size + 1))
        img[y:y + size, x:x + size] = _scale_patch(face, size)
        boxes.append((x, y, size, size))
    return img, boxes


def planted_bgr(rows, cols, seed, nfaces=8, min_size=40, max_size=None, skin=(150, 110)):
    """BGR 8UC3 frame for the FaceDetector::detectFaces path: the planted gray frame as luma, neutral
    chroma (Cr = Cb = 128, outside the default skin bounds) on the background and skin chroma
    (Cr, Cb) = `skin` inside the planted patches plus a few decoy skin-coloured blobs without a
    face.  Integer inverse of the BT.601 YCrCb transform; returns (bgr, boxes)."""
    gray, boxes = planted(rows, cols, seed, nfaces, min_size, max_size)
    cr = np.full((rows, cols), 128, np.int64)
    cb = np.full((rows, cols), 128, np.int64)
    for (x, y, w, h) in boxes:
        cr[y:y + h, x:x + w] = skin[0]
        cb[y:y + h, x:x + w] = skin[1]
    rnd = splitmix64(seed ^ 0x5C1, 9)
    for k in range(3):                                   # decoys: skin colour, no face
        w = 30 + int(rnd[3 * k] % np.uint64(60))
        x = int(rnd[3 * k + 1] % np.uint64(max(cols - w, 1)))
        y = int(rnd[3 * k + 2] % np.uint64(max(rows - w, 1)))
        cr[y:y + w, x:x + w] = skin[0]
        cb[y:y + w, x:x + w] = skin[1]
    Y = gray.astype(np.int64)
    r = Y + (1403 * (cr - 128)) // 1000
    g = Y - (714 * (cr - 128) + 344 * (cb - 128)) // 1000
    b = Y + (1773 * (cb - 128)) // 1000
    bgr = np.stack([b, g, r], axis=-1)
    return np.clip(bgr, 0, 255).astype(np.uint8), boxes


def make_frames(kind, n, rows, cols, first=0):
    out = np.zeros((n, rows, cols), np.uint8)
    for i in range(n):
        seed = frame_seed(first + i)
        if kind == "noise":
            out[i] = noise(rows, cols, seed)
        elif kind == "texture":
            out[i] = texture(rows, cols, seed)
        elif kind == "planted":
            out[i] = planted(rows, cols, seed)[0]
        else:
            raise ValueError(kind)
    return out
