#!/usr/bin/env python3
"""First Gauss-Newton step of a fuzz case: the oracle's solver (= the product's, bit for bit on the host) on HIP's and on the oracle's H, G."""
import ast, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import bpvo_amd
import __graft_entry__ as ge
from bpvo_amd import capi
from util import make_params
import fuzz_parity as fz
np.set_printoptions(linewidth=220, precision=5)
hip = bpvo_amd.load(); orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
lines = [l.strip() for l in open(os.path.join(ROOT, "tests/tools/fuzz_regressions.txt")) if l.strip() and not l.startswith("#")]
for which in (int(a) for a in sys.argv[1:]) if len(sys.argv) > 1 else range(len(lines)):
    head, brace = lines[which].split("{", 1)
    rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
    kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
    K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
    kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
    res = {}
    for name, bind in (("hip", hip), ("orc", orc)):
        c = bind.create(K, b, rows, cols, make_params(bind, **kw2), n_frames=2, n_pairs=1)
        c.frame_set_data(0, imgA, dispA); c.frame_set_data(1, imgB, dispB); c.frame_set_template(0)
        res[name] = c.linearize(0, 0, 1, kw2["levels"] - 1, np.eye(4, dtype=np.float32))
        c.close()
    print(f"case {which}: {rows}x{cols} {kw['descriptor']}/{kw['loss']} norm={kw['withNormalization']}  cond(H) %.2e  n_valid %d" % (np.linalg.cond(res["orc"]["H"].astype(np.float64)), res["orc"]["num_valid"]))
    for name in ("hip", "orc"):
        H = np.ascontiguousarray(res[name]["H"], np.float32); G = np.ascontiguousarray(res[name]["G"], np.float32)
        dp = np.zeros(6, np.float32)
        ok = orc.fn("solve")(H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p))
        dp64 = np.linalg.solve(H.astype(np.float64), G.astype(np.float64))
        resid = H.astype(np.float64) @ dp.astype(np.float64) - G
        print("  %s: solver ok=%d dp %s |dp| %.4f   f64 solve of the same system: |dp| %.4f   (H dp - G)^2 / min(...) = %.2e" % (
            name, ok, dp, np.linalg.norm(dp), np.linalg.norm(dp64), (resid @ resid) / min((H.astype(np.float64) @ dp) @ (H.astype(np.float64) @ dp), G.astype(np.float64) @ G)))
