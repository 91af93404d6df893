#!/usr/bin/env python3
"""For the cases of fuzz_regressions.txt: the HIP path against the oracle with its reference accumulation (f32, serial) AND
against the oracle with the same sums accumulated in f64 (bpvo_orc_set_reduction(1): a test instrument, not a mode of the
reference).  The GPU reduction (wave tree + f64 block combine) is within 4e-6 of the f64 sums, so where the termination of a
level is decided by the rounding of H, G, f_norm the HIP run should follow the f64 oracle, not the f32 one.
usage (GPU box): python tests/tools/replay_vs_f64.py [tests/tools/fuzz_regressions.txt]"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import numpy as np  # noqa: E402
import bpvo_amd  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from bpvo_amd import capi  # noqa: E402
from util import ROT_TOL, make_params, pose_error, trans_tol  # noqa: E402
import fuzz_parity as fz  # noqa: E402

hip = bpvo_amd.load()
orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "tools", "fuzz_regressions.txt")
for line in open(path):
    line = line.strip()
    if not line or line.startswith("#"):
        continue
    head, brace = line.split("{", 1)
    rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
    kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
    K, b, imgA, dispA, imgB, dispB, slack = fz.make_inputs(rows, cols, scene, seed)
    kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
    form = 2 if kw.get("_dspace") else (1 if kw.get("_fast_warp") else 0)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if kw.get("_fuse_frozen") else "0")
    out = {}
    for name, bind, red in (("hip", hip, 0), ("f32", orc, 0), ("f64", orc, 1)):
        ctx = bind.create(K, b, rows, cols, make_params(bind, **kw2), n_frames=2, n_pairs=1)
        if form:
            ctx.set_warp_formulation(form)
        if bind is orc:
            ctx.call("set_reduction", red)
        ctx.frame_set_data(0, imgA, dispA); ctx.frame_set_data(1, imgB, dispB); ctx.frame_set_template(0)
        T, st = ctx.estimate_pose(0, 0, 1)
        out[name] = (T, [(s["numIterations"], hex(s["status"])[2:]) for s in st])
        ctx.close()
    e32, e64 = pose_error(out["hip"][0], out["f32"][0]), pose_error(out["hip"][0], out["f64"][0])
    bar = (slack * ROT_TOL, slack * trans_tol(K))
    verdict = "within the bar of the f32 oracle" if e32[0] <= bar[0] and e32[1] <= bar[1] else (
        "follows the f64-accumulating oracle" if e64[0] <= bar[0] and e64[1] <= bar[1] else "neither")
    print(f"{rows}x{cols} seed {seed} {kw['descriptor']}/{kw['loss']}: hip vs f32 {e32[0]:.1e} rad {e32[1]:.1e} m | hip vs f64 {e64[0]:.1e} {e64[1]:.1e} | "
          f"bar {bar[0]:.1e} {bar[1]:.1e} | its hip {out['hip'][1]} f32 {out['f32'][1]} f64 {out['f64'][1]} -> {verdict}", flush=True)
