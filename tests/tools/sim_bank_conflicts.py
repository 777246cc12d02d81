#!/usr/bin/env python3
"""CPU model of the LDS bank conflicts of k_cascade's compacted stages ("What bounds
k_cascade").  Uses the oracle's per-window stage taps of two 720p 'planted' frames to rebuild, per tile
and stage, the survivor list the kernel would hold, and counts LDS cycles per 32-lane group
(max entries on one bank; bank = window word offset mod 32) for several list orders and tile pitches.
Test-side analysis only (it runs the oracle); nothing in the product path uses it.

  python tests/tools/sim_bank_conflicts.py      # a few minutes on one core
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest, oracle_lib as ol
conftest.load_package()
import importlib
synth = importlib.import_module('ugproj_detector_amd.synth')
c = ol.OracleCascade(ol.cascade_path('alt'))
sn = c.dump()[1]
ROWS=16
fr = synth.make_frames('planted', 2, 720, 1280, first=0)
levels,_ = c.plan(720,1280,1.1,(30,30),(0,0))
def sim(order, stride1, stride2, sparse_max=32):
    # returns dict: total weighted cycles and ideal
    cyc = 0.0; ideal = 0.0; sparse_feats=0; dense_feats=0; lanes_used=0; lanes_tot=0
    for f in fr:
        g, raw, taps = c.detect(f, 1.1, 2, (30,30), (0,0), want_taps=True)
        for L, t in zip(levels, taps):
            ys = L.ystep; stride = stride1 if ys==1 else stride2
            ny, nx = t.shape
            for gy0 in range(0, ny, ROWS):
                for gx0 in range(0, nx, 64):
                    tt = t[gy0:gy0+ROWS, gx0:gx0+64]
                    rr, xx = np.nonzero(tt >= 1)
                    if order == 'col':
                        o = np.lexsort((rr, xx)); rr, xx = rr[o], xx[o]
                    st = tt[rr, xx]
                    for s in range(1, 22):
                        m = st >= s
                        n = int(m.sum())
                        if n == 0: break
                        if n <= sparse_max:
                            # sparse finish: all remaining
                            for s2 in range(s,22):
                                sparse_feats += int((st>=s2).sum())*int(sn[s2])
                            break
                        r, x = rr[m], xx[m]
                        bank = (r*stride + x) % 32
                        w = int(sn[s])
                        dense_feats += n*w
                        for i in range(0, n, 32):
                            b = bank[i:i+32]
                            mx = np.bincount(b, minlength=32).max()
                            cyc += mx*w; ideal += w
                        lanes_used += n*w; lanes_tot += ((n+63)//64)*64*w
    return cyc/ideal, dense_feats, sparse_feats, lanes_used/lanes_tot
for order in ('row','col'):
  for s1,s2 in ((84,296),(85,297),(88,296),(92,300),(96,288),(97,289),(100,296),(68,0),(72,264),(76,268)):
    print(order, s1, s2, sim(order, s1, s2 if s2 else 296))
print("---- residue-bucket packing")
def sim2(stride1, stride2, sparse_max=32, nres=32):
    cur_passes=0; cur_cyc=0; b_passes=0; tot=0
    for f in fr:
        g, raw, taps = c.detect(f, 1.1, 2, (30,30), (0,0), want_taps=True)
        for L, t in zip(levels, taps):
            ys = L.ystep; stride = stride1 if ys==1 else stride2
            ny, nx = t.shape
            for gy0 in range(0, ny, ROWS):
                for gx0 in range(0, nx, 64):
                    tt = t[gy0:gy0+ROWS, gx0:gx0+64]
                    rr, xx = np.nonzero(tt >= 1)
                    st = tt[rr, xx]
                    for s in range(1, 22):
                        m = st >= s
                        n = int(m.sum())
                        if n == 0 or n <= sparse_max: break
                        r, x = rr[m], xx[m]
                        bank = (r*stride + x) % nres
                        w = int(sn[s])
                        k = np.bincount(bank, minlength=nres).max()
                        groups = max((n+31)//32, k)          # 32-lane groups needed
                        b_passes += ((groups+1)//2)*w        # wave passes (2 groups per wave)
                        cur_passes += ((n+63)//64)*w
                        tot += n*w
    return cur_passes, b_passes, b_passes/cur_passes
for s1,s2 in ((84,296),(96,320),(97,321),(100,328),(104,336),(88,304),(92,312)):
    print(s1,s2,sim2(s1,s2))
print("---- hybrid: strict layers while fill >= F, dense tail")
def sim3(F, stride1=84, stride2=296, sparse_max=32):
    cur_groups=0; cur_cyc=0; h_groups=0; h_cyc=0
    for f in fr:
        g, raw, taps = c.detect(f, 1.1, 2, (30,30), (0,0), want_taps=True)
        for L, t in zip(levels, taps):
            ys = L.ystep; stride = stride1 if ys==1 else stride2
            ny, nx = t.shape
            for gy0 in range(0, ny, ROWS):
                for gx0 in range(0, nx, 64):
                    tt = t[gy0:gy0+ROWS, gx0:gx0+64]
                    rr, xx = np.nonzero(tt >= 1)
                    st = tt[rr, xx]
                    for s in range(1, 22):
                        m = st >= s
                        n = int(m.sum())
                        if n == 0 or n <= sparse_max: break
                        r, x = rr[m], xx[m]
                        bank = (r*stride + x) % 32
                        w = int(sn[s])
                        # current
                        for i in range(0, n, 32):
                            cur_cyc += np.bincount(bank[i:i+32], minlength=32).max()*w
                            cur_groups += w
                        k = np.bincount(bank, minlength=32)
                        J = 0
                        while (k > J).sum() >= F: J += 1
                        # strict layers
                        h_groups += J*w; h_cyc += J*w
                        # tail: entries beyond J per residue, in residue-major order
                        tail = np.repeat(np.arange(32), np.maximum(k - J, 0))
                        # interleave to spread residues: order by (layer, residue)
                        lay = np.concatenate([np.arange(max(kk-J,0)) for kk in k]) if len(tail) else np.array([],int)
                        o = np.lexsort((tail, lay)); tail = tail[o]
                        for i in range(0, len(tail), 32):
                            h_cyc += np.bincount(tail[i:i+32], minlength=32).max()*w
                            h_groups += w
    return cur_groups, cur_cyc, h_groups, h_cyc, h_groups/cur_groups, h_cyc/cur_cyc
for F in (32, 28, 24, 20, 16, 1):
    print(F, sim3(F))
