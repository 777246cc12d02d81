"""Mean weak classifiers evaluated per window on one synthetic 1280x720 frame (CPU oracle stage taps), for DESIGN.md section 5:
    python tools/feature_stats.py [planted|noise|texture] [alt|alt2|default|alt_tree]"""
import sys, os, numpy as np
sys.path.insert(0, 'tests')
import conftest
pkg = conftest.load_package(); synth = conftest.load_synth()
import oracle_lib; oracle_lib.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else 'planted'
name = sys.argv[2] if len(sys.argv) > 2 else 'alt'
fr = synth.make_frames(kind, 1, 720, 1280, first=3)[0]
path = 'ugproj-detector_amd/data/cascades/haarcascade_frontalface_%s.hcc' % name
oc = oracle_lib.OracleCascade(path)
g, raw, taps = oc.detect(fr, 1.1, 2, (30, 30), (0, 0), want_taps=True)
sys.path.insert(0, 'oracle'); import np_oracle
c = np_oracle.load_hcc(path)
# stage sizes
sizes = [int(x) for x in c.stage_ntrees]
print('stage sizes', sizes[:8])
cum = np.concatenate([[0], np.cumsum(sizes)])
tot_w = tot_f = 0
hist = np.zeros(len(sizes) + 2, np.int64)
for li, t in enumerate(taps):
    t = np.asarray(t).astype(int)
    v = t[t >= 0] if (t < 0).any() else t.ravel()
    # taps: stages passed; evaluated stages = passed + 1 (capped)
    ev = np.minimum(np.abs(v) + 1, len(sizes))
    tot_w += v.size; tot_f += cum[ev].sum()
    hist += np.bincount(np.minimum(np.abs(v), len(sizes)), minlength=len(hist))[:len(hist)]
print('windows', tot_w, 'features/window', tot_f / tot_w)
print('passed-stage histogram (fraction):', np.round(hist / hist.sum(), 4)[:10])
