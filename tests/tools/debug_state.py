#!/usr/bin/env python3
import ast, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import bpvo_amd
import __graft_entry__ as ge
from bpvo_amd import capi
from util import make_params
import fuzz_parity as fz
np.set_printoptions(linewidth=220, precision=6, suppress=True)
hip = bpvo_amd.load(); orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
lines = [l.strip() for l in open(os.path.join(ROOT, "tests/tools/fuzz_regressions.txt")) if l.strip() and not l.startswith("#")]
which = int(sys.argv[1]) if len(sys.argv) > 1 else 8
head, brace = lines[which].split("{", 1)
rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
kw2["maxTestLevel"] = kw2["levels"] - 1
kw2["maxIterations"] = 0
ch = hip.create(K, b, rows, cols, make_params(hip, **kw2), n_frames=2, n_pairs=1)
co = orc.create(K, b, rows, cols, make_params(orc, **kw2), n_frames=2, n_pairs=1)
for c in (ch, co):
    c.frame_set_data(0, imgA, dispA); c.frame_set_data(1, imgB, dispB); c.frame_set_template(0)
Th, sh = ch.estimate_pose(0, 0, 1)
To, so, tr = co.estimate_pose_trace(0, 0, 1)
st = np.zeros(84, np.float32); ch.call("debug_gn_state", 0, st.ctypes.data_as(C.c_void_p))
print("oracle trace records:", len(tr))
for k, rec in enumerate(tr):
    print("orc lin %d: T row0 %s t %s | dp %s f %.6f sigma %.6f" % (k, rec[:3], rec[[3, 7, 11]], rec[61:67], rec[58], rec[59]))
print("hip T_lin row0 %s t %s | dp %s f %.6f sigma %.6f" % (st[68:71], st[[71, 75, 79]], st[58:64], st[64], st[65]))
print("hip final T t", Th[:3, 3], " orc final T t", To[:3, 3])
lv = kw2["levels"] - 1
a = ch.linearize(0, 0, 1, lv, np.eye(4, dtype=np.float32)); bb = co.linearize(0, 0, 1, lv, np.eye(4, dtype=np.float32))
print("linearize at I: G hip", a["G"], "G orc", bb["G"])
nT, nTi = ch.get_normalization(0, lv)
print("normalisation hip", nT.reshape(-1)[[0, 3, 7, 11]], " orc", co.get_normalization(0, lv)[0].reshape(-1)[[0, 3, 7, 11]])
# second linearisation: the oracle's solver on HIP's and on the oracle's H, G at the oracle's pose; sensitivity of its fallback decision
rec = tr[1]
T1 = rec[:16].reshape(4, 4)
a = ch.linearize(0, 0, 1, lv, T1, reset_scale=False)
Ho, Go = np.ascontiguousarray(rec[16:52].reshape(6, 6), np.float32), np.ascontiguousarray(rec[52:58], np.float32)
def solve(H, G):
    H = np.ascontiguousarray(H, np.float32); G = np.ascontiguousarray(G, np.float32); dp = np.zeros(6, np.float32)
    orc.fn("solve")(H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p))
    return dp
print("second linearisation: cond(H) %.2e" % np.linalg.cond(Ho.astype(np.float64)))
print("  solver on the oracle's H, G: |dp| %.4f" % np.linalg.norm(solve(Ho, Go)), " on HIP's H, G (sigma %.6f vs %.6f): |dp| %.4f" % (a["sigma"], rec[59], np.linalg.norm(solve(a["H"], a["G"]))))
rng = np.random.default_rng(0)
big = 0
for k in range(200):
    E = 1.0 + 2e-7 * rng.standard_normal((6, 6)); E = (E + E.T) / 2
    big += np.linalg.norm(solve(Ho * E, Go * (1.0 + 2e-7 * rng.standard_normal(6)))) > 0.1
print("  200 relative perturbations of 2e-7 of the oracle's system: the undamped step (|dp| > 0.1) in %d, the damped fallback in %d" % (big, 200 - big))
