#!/usr/bin/env python3
"""Entry-wise accuracy of H and G at a converged pose: HIP vs the oracle (f32 serial) vs an f64 evaluation of the same J, r, w."""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import bpvo_amd
import __graft_entry__ as ge
from bpvo_amd import capi
from util import make_params, pose_error
import fuzz_parity as fz
np.set_printoptions(linewidth=220, precision=3)
hip = bpvo_amd.load(); orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
which = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lines = [l.strip() for l in open(os.path.join(ROOT, "tests/tools/fuzz_regressions.txt")) if l.strip() and not l.startswith("#")]
head, brace = lines[which].split("{", 1)
rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
os.environ["BPVO_HIP_FUSE_FROZEN"] = "1" if kw.get("_fuse_frozen") else "0"
ctx = {}
for name, bind in (("hip", hip), ("orc", orc)):
    c = bind.create(K, b, rows, cols, make_params(bind, **kw2), n_frames=2, n_pairs=1)
    c.frame_set_data(0, imgA, dispA); c.frame_set_data(1, imgB, dispB); c.frame_set_template(0)
    ctx[name] = c
To, so, trace = ctx["orc"].estimate_pose_trace(0, 0, 1)
Th, sh = ctx["hip"].estimate_pose(0, 0, 1)
print(kw["descriptor"], kw["loss"], rows, cols, "orc its", [(s["numIterations"], hex(s["status"])) for s in so], "hip its", [(s["numIterations"], hex(s["status"])) for s in sh], "pose err", pose_error(Th, To))
l = kw2.get("maxTestLevel", 0)
for label, T in (("oracle final pose", To), ("hip final pose", Th)):
    a = ctx["hip"].linearize(0, 0, 1, l, T); bb = ctx["orc"].linearize(0, 0, 1, l, T)
    J = ctx["orc"].get_jacobians(0, l).astype(np.float64).reshape(-1, 6); r = ctx["orc"].get_residuals(0).astype(np.float64)
    w = ctx["orc"].get_weights(0).astype(np.float64); v = np.tile(ctx["orc"].get_valid(0).astype(np.float64), ctx["orc"].Cn)
    wv = w * v
    H64 = (J * wv[:, None]).T @ J; G64 = J.T @ (wv * r)
    Habs = (np.abs(J) * wv[:, None]).T @ np.abs(J); Gabs = np.abs(J).T @ (wv * np.abs(r))
    print("==", label, "sigma hip/orc", a["sigma"], bb["sigma"], "n", len(r), "cond(H64) %.2e" % np.linalg.cond(H64))
    print("G64      ", G64)
    print("G hip err / sum|terms|", (a["G"] - G64) / Gabs)
    print("G orc err / sum|terms|", (bb["G"] - G64) / Gabs)
    print("H hip err / sum|terms| max", np.abs((a["H"] - H64) / Habs).max(), " orc", np.abs((bb["H"] - H64) / Habs).max())
    dp64 = np.linalg.solve(H64, G64)
    dph = np.linalg.solve(a["H"].astype(np.float64), a["G"].astype(np.float64)); dpo = np.linalg.solve(bb["H"].astype(np.float64), bb["G"].astype(np.float64))
    print("dp f64", dp64, "|dp|", np.linalg.norm(dp64)); print("dp hip", dph, "|dp|", np.linalg.norm(dph)); print("dp orc", dpo, "|dp|", np.linalg.norm(dpo))
