#!/usr/bin/env python3
"""Debug aid: replay the oracle's pose sequence of one fuzz case on the HIP path, linearisation by linearisation."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bpvo_amd, __graft_entry__ as ge
from bpvo_amd import capi
from util import make_params, bits_equal
import fuzz_parity as fz

rows, cols, scene, seed = 67, 297, 0, 356139139
kw = {'descriptor': 'centraldiff', 'loss': 'huber', 'levels': 2, 'gradientEstimation': 1, 'withNormalization': 0, 'interp': 1, 'minNumPixelsForNonMaximaSuppression': 1000000000, 'nonMaxSuppRadius': 1, 'minSaliency': 1.0, 'maxTestLevel': 0, 'centralDifferenceRadius': 1, 'centralDifferenceSigmaBefore': -1.0, 'centralDifferenceSigmaAfter': 1.75}
hip = bpvo_amd.load(); orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
ctxs = []
for bind in (hip, orc):
    ctx = bind.create(K, b, rows, cols, make_params(bind, **kw), n_frames=2, n_pairs=1)
    ctx.frame_set_data(0, imgA, dispA); ctx.frame_set_data(1, imgB, dispB); ctx.frame_set_template(0)
    ctxs.append(ctx)
ch, co = ctxs
print("points", ch.num_points(0, 0), "C", ch.Cn)
To, so, rec = co.estimate_pose_trace(0, 0, 1)
print("oracle stats", so, "records", len(rec))
Th, sh = ch.estimate_pose(0, 0, 1)
print("hip stats", sh)
lvl = [int(r[67]) for r in rec]
print("levels of records", lvl)
first0 = lvl.index(0)
T_start = rec[first0][:16].reshape(4, 4).copy()
# GPU's own iteration at level 0 from the oracle's start pose of that level: emulate GN by hand is not possible through the
# ABI, so compare the linearisations along the oracle's path
for k, r in enumerate(rec[first0:]):
    T = r[:16].reshape(4, 4); H = r[16:52].reshape(6, 6); G = r[52:58]; fn, sig, nv = r[58], r[59], r[60]
    a = ch.linearize(0, 0, 1, 0, T, reset_scale=(k == 0))
    b2 = co.linearize(0, 0, 1, 0, T, reset_scale=(k == 0))
    same_r = bits_equal(ch.get_residuals(0), co.get_residuals(0))
    same_w = bits_equal(ch.get_weights(0), co.get_weights(0))
    dH = np.abs(a["H"] - H).max() / np.abs(H).max(); dG = np.abs(a["G"] - G).max() / max(np.abs(G).max(), 1e-30)
    dG2 = np.abs(a["G"] - b2["G"]).max() / max(np.abs(G).max(), 1e-30)
    print(k, "sigma hip %.9g orc-trace %.9g orc-lin %.9g" % (a["sigma"], sig, b2["sigma"]), "r", same_r, "w", same_w, "dH %.2e dG %.2e dG(lin) %.2e |G| %.3e f %.6g/%.6g nv %d/%d" % (dH, dG, dG2, np.abs(G).max(), a["f_norm"], fn, a["num_valid"], nv))
    if k > 40:
        break
