#!/usr/bin/env python3
"""Where does an estimate loop of the HIP path leave the oracle's?  (1) HIP linearised at every pose of the oracle's trace
(scale estimator history kept), (2) the level-by-level result of HIP's own loop."""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import bpvo_amd
import __graft_entry__ as ge
from bpvo_amd import capi
from util import make_params, pose_error
import fuzz_parity as fz
np.set_printoptions(linewidth=220, precision=4)
hip = bpvo_amd.load(); orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
which = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
print(kw, rows, cols)
prev_level = -1
for k, rec in enumerate(trace):
    level = int(rec[67]); T = rec[:16].reshape(4, 4)
    a = ctx["hip"].linearize(0, 0, 1, level, T, reset_scale=(level != prev_level))
    prev_level = level
    Ho, Go = rec[16:52].reshape(6, 6), rec[52:58]
    print("it %2d L%d  sigma hip %.7g orc %.7g  nvalid %d/%d  f hip %.6f orc %.6f  |dH|/|H| %.1e |dG|/|G| %.1e  |dp_orc| %.2e" % (
        k, level, a["sigma"], rec[59], a["num_valid"], int(rec[60]), a["f_norm"], rec[58], np.abs(a["H"] - Ho).max() / np.abs(Ho).max(),
        np.abs(a["G"] - Go).max() / max(np.abs(Go).max(), 1e-30), np.linalg.norm(rec[61:67])))
# HIP's own loop, one level at a time through maxTestLevel (estimate_pose runs levels L-1 .. maxTestLevel)
Th, sh = ctx["hip"].estimate_pose(0, 0, 1)
print("oracle:", [(s["numIterations"], hex(s["status"]), round(s["finalError"], 5)) for s in so])
print("hip   :", [(s["numIterations"], hex(s["status"]), round(s["finalError"], 5)) for s in sh], "pose err", pose_error(Th, To))
