#!/usr/bin/env python3
"""Writes tests/golden/oracle_vectors.npz: small committed frames + the CPU oracle's outputs on
them for all four reference cascades (call parameters of reference src/tracker/detector.cc:25-31:
scaleFactor 1.1, minNeighbors 2, minSize 30x30).

These are REGRESSION vectors of the oracle (parity unpinned: no OpenCV, no reference fixtures --
SURVEY.md 8c).  They are data: inputs and expected outputs; no reference source is involved.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import oracle_lib  # noqa: E402

synth = conftest.load_synth()
frames = np.stack([
    synth.planted(120, 160, synth.frame_seed(100), nfaces=2, min_size=40, max_size=90)[0],
    synth.planted(120, 160, synth.frame_seed(101), nfaces=3, min_size=36, max_size=70)[0],
    synth.texture(120, 160, synth.frame_seed(102)),
])
out = {"frames": frames}
for name in oracle_lib.CASCADE_NAMES:
    c = oracle_lib.OracleCascade(oracle_lib.cascade_path(name))
    for i, fr in enumerate(frames):
        g, raw, taps = c.detect(fr, 1.1, 2, (30, 30), (0, 0), want_taps=True)
        out["%s_%d_grouped" % (name, i)] = g
        out["%s_%d_raw" % (name, i)] = raw
        out["%s_%d_taphist" % (name, i)] = np.bincount(
            np.concatenate([t.ravel() for t in taps]).astype(np.int64) + 2, minlength=52)
        print(name, i, "grouped", len(g), "raw", len(raw))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"), **out)
