#!/usr/bin/env python3
"""Pack an old-format OpenCV Haar cascade XML (reference data/haarcascade_frontalface_*.xml,
root `type_id="opencv-haar-classifier"`, e.g. data/haarcascade_frontalface_alt.xml:46) into the
compact binary "HCC1" container used as a travelling fixture.

Why: /root/reference does not exist on the GPU box and reference files must not be copied
into this repository; the model *parameters* (numbers, not text) are re-encoded here.
The product loader (ugproj-detector_amd/csrc/cascade_model.cpp) reads both the XML and HCC1;
tests check that both give identical models when the XML is reachable.

Numeric conversion follows SURVEY.md A.2: decimal text -> IEEE f64 (Python float == strtod)
-> f32 (numpy RNE) for node thresholds, leaf values and rect weights; stage thresholds are
stored as f64 (the loader applies `(float)t - 1e-5f`).

Layout (little endian):
  "HCC1" i32 win_w win_h nstages ntrees nnodes nleaves flags(bit0 = any tilted)
  nstages x { f64 threshold, i32 ntrees, i32 parent, i32 next }
  ntrees  x { i32 nnodes }
  nnodes  x { i8 rect[3][4] (x,y,w,h), f32 weight[3], f32 threshold, i32 left, i32 right, i32 tilted }
           left/right: <=0 -> -(leaf index inside the tree), >0 -> node index inside the tree
  nleaves x { f32 }
"""
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np


def parse_xml(path):
    root = ET.parse(path).getroot()
    top = None
    for ch in root:
        if ch.attrib.get("type_id") == "opencv-haar-classifier":
            top = ch
            break
    if top is None:
        raise ValueError("no opencv-haar-classifier node in %s" % path)
    win_w, win_h = [int(t) for t in top.find("size").text.split()]
    stages, trees, nodes, leaves = [], [], [], []
    tilted_any = 0
    for st in top.find("stages"):
        tr = st.find("trees")
        par = st.find("parent")
        nxt = st.find("next")
        stages.append((float(st.find("stage_threshold").text), len(tr),
                       int(par.text) if par is not None else -1,
                       int(nxt.text) if nxt is not None else -1))
        for tree in tr:
            trees.append(len(tree))
            nleaf = 0
            for nd in tree:
                fe = nd.find("feature")
                rects = []
                for r in fe.find("rects"):
                    t = r.text.split()
                    rects.append(([int(v) for v in t[:4]], np.float32(float(t[4]))))
                while len(rects) < 3:
                    rects.append(([0, 0, 0, 0], np.float32(0)))
                tl = int(fe.find("tilted").text)
                tilted_any |= tl
                thr = np.float32(float(nd.find("threshold").text))
                lv = nd.find("left_val")
                if lv is not None:
                    left = -nleaf
                    leaves.append(np.float32(float(lv.text)))
                    nleaf += 1
                else:
                    left = int(nd.find("left_node").text)
                rv = nd.find("right_val")
                if rv is not None:
                    right = -nleaf
                    leaves.append(np.float32(float(rv.text)))
                    nleaf += 1
                else:
                    right = int(nd.find("right_node").text)
                nodes.append((rects, thr, left, right, tl))
            assert nleaf == len(tree) + 1
    return dict(win_w=win_w, win_h=win_h, stages=stages, trees=trees, nodes=nodes,
                leaves=leaves, tilted=tilted_any)


def pack(model):
    out = bytearray(b"HCC1")
    out += struct.pack("<7i", model["win_w"], model["win_h"], len(model["stages"]), len(model["trees"]),
                       len(model["nodes"]), len(model["leaves"]), model["tilted"])
    for thr, nt, par, nxt in model["stages"]:
        out += struct.pack("<d3i", thr, nt, par, nxt)
    for n in model["trees"]:
        out += struct.pack("<i", n)
    for rects, thr, left, right, tl in model["nodes"]:
        for r, _ in rects:
            out += struct.pack("<4b", *r)
        for _, w in rects:
            out += struct.pack("<f", float(w))
        out += struct.pack("<f3i", float(thr), left, right, tl)
    for v in model["leaves"]:
        out += struct.pack("<f", float(v))
    return bytes(out)


def main(argv):
    if len(argv) != 3:
        print("usage: pack_cascade.py in.xml out.hcc", file=sys.stderr)
        return 2
    data = pack(parse_xml(argv[1]))
    with open(argv[2], "wb") as f:
        f.write(data)
    print("%s -> %s (%d bytes)" % (argv[1], argv[2], len(data)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
