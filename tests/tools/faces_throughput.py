#!/usr/bin/env python3
"""Batched FaceDetector::detectFaces (BGR -> skin mask, masked equalised gray, the cascade call, skin-ratio filter)
as a two-slot software pipeline on one stream (test-side measurement: it checks against the CPU oracle); device-resident BGR frames, BASELINE config #1's scan size (640x360) and 720p.
Checks the first batch against the CPU oracle's restatement of detectFaces, then prints frames/s.  Run on the GPU box."""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import conftest
pkg = conftest.load_package()
import oracle_lib
synth = importlib.import_module("ugproj_detector_amd.synth")
lib = pkg.lib


def run(rows, cols, B, steps, name="alt2"):
    frames = np.stack([synth.planted_bgr(rows, cols, synth.frame_seed(4000 + i), nfaces=4, min_size=40, max_size=min(rows, cols) // 2)[0]
                       for i in range(min(B, 32))])
    frames = frames[np.arange(B) % len(frames)]
    d = torch.from_numpy(frames).cuda()
    stream = torch.cuda.Stream()
    casc = pkg.Cascade(oracle_lib.cascade_path(name))
    lo, up = (C.c_int * 3)(*pkg.SKIN_LOWER), (C.c_int * 3)(*pkg.SKIN_UPPER)
    slots = []
    for _ in range(2):
        c = pkg.Context(casc, device=0, stream=stream.cuda_stream)
        c.configure(rows, cols, 1.1, 2, (30, 30), (0, 0), max_batch=B, cand_capacity=4096)
        c.enable_faces()
        slots.append(c)
    out = np.zeros((1 << 16, 4), np.int32)
    ofs = np.zeros(B + 1, np.int32)

    def enqueue(c):
        pkg._check(lib.hc_faces_batch_enqueue(c._h, d.data_ptr(), rows * cols * 3, cols * 3, B, lo, up))

    def wait(c):
        pkg._check(lib.hc_faces_batch_wait(c._h, out.ctypes.data, len(out), ofs.ctypes.data))
        return [out[ofs[i]:ofs[i + 1]].copy() for i in range(B)]

    enqueue(slots[0])
    first = wait(slots[0])
    oc = oracle_lib.OracleCascade(oracle_lib.cascade_path(name))
    nb = 0
    for f in range(4):
        want = oc.detect_faces(frames[f], pkg.SKIN_LOWER, pkg.SKIN_UPPER, 1.1)
        assert np.array_equal(oracle_lib.sorted_boxes(first[f]), oracle_lib.sorted_boxes(want)), (f, first[f], want)
        nb += len(want)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enqueue(slots[0])
        for i in range(steps):
            if i + 1 < steps:
                enqueue(slots[(i + 1) & 1])
            got = wait(slots[i & 1])
        dt = time.perf_counter() - t0
    assert all(np.array_equal(a, b) for a, b in zip(got, first))
    print("detectFaces %dx%d BGR, %s, batches of %d: %.0f frames/s (%.2f ms per batch), %d boxes in the 4 checked frames, bit-exact vs the oracle"
          % (cols, rows, name, B, B * steps / dt, 1e3 * dt / steps, nb))
    for c in slots:
        c.close()


if __name__ == "__main__":
    run(360, 640, 256, 12)
    run(720, 1280, 128, 8)
