#!/usr/bin/env python3
"""Both sides' Gauss-Newton traces of ONE case of the randomised parity tool, side by side at the finest level: pose distance, f_norm,
sigma, |dp| per linearisation — where the two runs part and on which of testConvergence's tests each ends.

  python tests/tools/diagnose_case.py 61 213 0 898244736 "{'descriptor': 'intensity', ...}"
"""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import bpvo_amd
    from bpvo_amd import capi
    import __graft_entry__ as ge
    import fuzz_parity as fz
    rows, cols, scene, seed = (int(v) for v in sys.argv[1:5])
    kw = ast.literal_eval(sys.argv[5])
    hip = bpvo_amd.load()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
    kw = dict(kw)
    fast_warp, fuse = kw.pop("_fast_warp", False), kw.pop("_fuse_frozen", False)
    formulation = 2 if kw.pop("_dspace", False) else (1 if fast_warp else 0)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0")
    ctxs = []
    for bind in (hip, orc):
        ctx = bind.create(K, b, rows, cols, fz.make_params(bind, **kw), n_frames=2, n_pairs=1)
        if formulation:
            ctx.set_warp_formulation(formulation)
        ctx.frame_set_data(0, imgA, dispA)
        ctx.frame_set_data(1, imgB, dispB)
        ctx.frame_set_template(0)
        ctxs.append(ctx)
    ch, co = ctxs
    Th, sh, trh = ch.estimate_pose_trace(0, 0, 1)
    To, so, tro = co.estimate_pose_trace(0, 0, 1)
    print("GPU   :", [(s["status"], s["numIterations"]) for s in sh], "oracle:", [(s["status"], s["numIterations"]) for s in so])
    first = kw.get("maxTestLevel", 0)
    for lvl in sorted(set(trh[:, 67].astype(int)) | set(tro[:, 67].astype(int)), reverse=True):
        h, o = trh[trh[:, 67] == lvl], tro[tro[:, 67] == lvl]
        print("level", lvl, "records GPU", len(h), "oracle", len(o), "points", ch.num_points(0, lvl))
        n = max(len(h), len(o))
        for i in range(n):
            a = h[i] if i < len(h) else None
            c = o[i] if i < len(o) else None
            line = "%4d" % i
            for t in (a, c):
                line += "  | " + ("f %.7g sigma %.7g |dp| %.3e |G|inf %.3e" % (t[58], t[59], float(np.linalg.norm(t[61:67])), float(np.abs(t[52:58]).max())) if t is not None else " " * 60)
            if a is not None and c is not None:
                r_, t_ = fz.pose_error(a[:16].reshape(4, 4), c[:16].reshape(4, 4))
                line += "  | apart %.2e rad %.2e m" % (r_, t_)
            if i < 12 or i > n - 12 or (a is None) != (c is None) or i % 25 == 0:
                print(line)
    for ctx in ctxs:
        ctx.close()


if __name__ == "__main__":
    main()
