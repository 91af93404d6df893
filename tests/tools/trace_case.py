#!/usr/bin/env python3
"""Side-by-side Gauss-Newton traces of one fuzz case (a line of a fuzz log / of fuzz_regressions.txt): f_norm per linearisation of the finest
estimated level on the HIP path, on the oracle and on the oracle with f64 accumulation, the distance of the HIP iterate from the oracle's,
and where the HIP pose lies relative to the oracle's iterates.   usage (GPU box): python tests/tools/trace_case.py <file> <rows> <cols>"""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bpvo_amd  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from bpvo_amd import capi  # noqa: E402
from util import make_params, pose_error  # noqa: E402
import fuzz_parity as fz  # noqa: E402


def main():
    path, want = sys.argv[1], " ".join(sys.argv[2:4])
    hip = bpvo_amd.load()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    for line in open(path):
        line = line.strip()
        if line.startswith("FAIL "):
            line = line[5:]
        if not line.startswith(want + " "):
            continue
        head, brace = line.split("{", 1)
        rows, cols, scene, seed = (int(v) for v in head.split()[:4])
        kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
        K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
        fast_warp, fuse = kw.pop("_fast_warp", False), kw.pop("_fuse_frozen", False)
        formulation = 2 if kw.pop("_dspace", False) else (1 if fast_warp else 0)
        os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0")
        print(rows, cols, kw, "formulation", formulation, "fuse", fuse, "slack", slack, "fx", K[0][0])
        cs = []
        for bind in (hip, orc):
            ctx = bind.create(K, b, rows, cols, make_params(bind, **kw), n_frames=2, n_pairs=1)
            if formulation:
                ctx.set_warp_formulation(formulation)
            ctx.frame_set_data(0, imgA, dispA); ctx.frame_set_data(1, imgB, dispB); ctx.frame_set_template(0)
            cs.append(ctx)
        ch, co = cs
        Th, sh, trh = ch.estimate_pose_trace(0, 0, 1)
        To, so, tro = co.estimate_pose_trace(0, 0, 1)
        co.call("set_reduction", 1); T64, s64, tr64 = co.estimate_pose_trace(0, 0, 1); co.call("set_reduction", 0)
        print("hip", [(s["status"], s["numIterations"]) for s in sh]); print("orc", [(s["status"], s["numIterations"]) for s in so])
        print("o64", [(s["status"], s["numIterations"]) for s in s64])
        print("final hip-orc", pose_error(Th, To), "hip-o64", pose_error(Th, T64), "orc-o64", pose_error(To, T64))
        lv = kw.get("maxTestLevel", 0)
        h0, o0, f0 = trh[trh[:, 67] == lv], tro[tro[:, 67] == lv], tr64[tr64[:, 67] == lv]
        print("k   f_hip  f_orc  f_o64   sigma_hip sigma_orc   d(hip,orc) d(orc,o64)")
        for k in range(max(len(h0), len(o0), len(f0))):
            row = ["%3d" % k]
            for t in (h0, o0, f0):
                row.append("%.6f" % t[k, 58] if k < len(t) else "-")
            row.append("%.6g" % h0[k, 59] if k < len(h0) else "-"); row.append("%.6g" % o0[k, 59] if k < len(o0) else "-")
            if k < len(h0) and k < len(o0):
                row.append("%.2e/%.2e" % pose_error(h0[k, :16].reshape(4, 4), o0[k, :16].reshape(4, 4)))
            if k < len(f0) and k < len(o0):
                row.append("%.2e/%.2e" % pose_error(f0[k, :16].reshape(4, 4), o0[k, :16].reshape(4, 4)))
            print(*row)
        for name, t in (("orc", o0), ("o64", f0)):
            d = [pose_error(Th, t[k, :16].reshape(4, 4))[0] for k in range(len(t))]
            print("HIP final pose closest to", name, "iterate", int(np.argmin(d)), "at %.2e rad" % min(d))


if __name__ == "__main__":
    main()
