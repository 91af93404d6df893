#!/usr/bin/env python3
"""Final pose / statistics of HIP and oracle as a function of maxIterations for one fuzz case: where do the loops part?"""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np
import bpvo_amd
import __graft_entry__ as ge
from bpvo_amd import capi
from util import make_params, pose_error
import fuzz_parity as fz
hip = bpvo_amd.load(); orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
which = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lines = [l.strip() for l in open(os.path.join(ROOT, "tests/tools/fuzz_regressions.txt")) if l.strip() and not l.startswith("#")]
head, brace = lines[which].split("{", 1)
rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
os.environ["BPVO_HIP_FUSE_FROZEN"] = "1" if kw.get("_fuse_frozen") else "0"
kw2["maxTestLevel"] = kw2["levels"] - 1     # the coarsest level only
for mi in list(range(0, 12)) + [20, 30, 50]:
    out = {}
    for name, bind in (("hip", hip), ("orc", orc)):
        c = bind.create(K, b, rows, cols, make_params(bind, **dict(kw2, maxIterations=mi)), n_frames=2, n_pairs=1)
        c.frame_set_data(0, imgA, dispA); c.frame_set_data(1, imgB, dispB); c.frame_set_template(0)
        T, st = c.estimate_pose(0, 0, 1)
        out[name] = (T, st[-1])
        c.close()
    e = pose_error(out["hip"][0], out["orc"][0])
    print("maxIterations %2d: pose diff %.2e rad %.2e m | hip %s | orc %s" % (mi, e[0], e[1], (out["hip"][1]["numIterations"], hex(out["hip"][1]["status"]), round(out["hip"][1]["finalError"], 6)),
          (out["orc"][1]["numIterations"], hex(out["orc"][1]["status"]), round(out["orc"][1]["finalError"], 6))))
