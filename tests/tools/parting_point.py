#!/usr/bin/env python3
"""Where, and on which test of PoseEstimatorBase::testConvergence (bpvo/pose_estimator_base.h:258-282), two Gauss-Newton runs of the same frame
pair part: the library in its default mode against the oracle — with the oracle's other summation orders (8-chunk = the reference's TBB build,
f64 accumulation) and the library's reference-order mode beside them.

Default case: conf/tsukuba_eval.cfg / Latch, frame 1 of the sequence of tests/test_gpu_reference_configs.py (coarsest level
81 iterations on the GPU against 3 on the oracle).  usage (GPU box):

    python tests/tools/parting_point.py [config] [descriptor] [frame]      # e.g.  tsukuba_eval latch 1
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bpvo_amd  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from bpvo_amd import capi, synth  # noqa: E402
from util import pose_error  # noqa: E402
import test_gpu_reference_configs as rc  # noqa: E402

SQRT_EPS = np.sqrt(np.float32(np.finfo(np.float32).eps))
STATUS = {capi.STATUS_MAX_ITERATIONS: "MaxIterations", capi.STATUS_FUNCTION_TOL: "FunctionTol", capi.STATUS_PARAMETER_TOL: "ParameterTol",
          capi.STATUS_GRADIENT_TOL: "GradientTol", capi.STATUS_SOLVER_ERROR: "SolverError"}


def tests_at(t, k, p_tol, f_tol, g_tol_param):
    """the three tests of testConvergence as they are evaluated BEFORE linearisation k + 1 of a level, from the trace records 0 .. k of that
    level: (dp_norm, dp_norm_prev, f, f_prev, g_norm, g_tol) and which one fires (None: the loop goes on)"""
    dp = np.float32(np.sqrt(np.sum(t[k, 61:67].astype(np.float32) ** 2, dtype=np.float32)))
    dp_prev = np.float32(0) if k == 0 else np.float32(np.sqrt(np.sum(t[k - 1, 61:67].astype(np.float32) ** 2, dtype=np.float32)))
    f, f_prev = np.float32(t[k, 58]), (np.float32(0) if k == 0 else np.float32(t[k - 1, 58]))
    g = np.float32(np.abs(t[k, 52:58]).max())
    g_tol = np.float32(g_tol_param) * max(np.float32(np.abs(t[0, 52:58]).max()), SQRT_EPS)
    fired = None
    if dp < p_tol or dp < np.float32(p_tol) * (SQRT_EPS + dp_prev):
        fired = "ParameterTol (:262-265)"
    elif f < f_tol or f < np.float32(f_tol) * (SQRT_EPS + f_prev) or abs(f - f_prev) < f_tol:
        fired = "FunctionTol (:267-271): |f - f_prev| = %.3g" % abs(float(f) - float(f_prev))
    elif g < g_tol:
        fired = "GradientTol (:273-276)"
    return dict(dp=float(dp), f=float(f), f_prev=float(f_prev), g=float(g), g_tol=float(g_tol), fired=fired)


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "tsukuba_eval"
    desc = sys.argv[2] if len(sys.argv) > 2 else "latch"
    frame = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows, cols, n = 480, 640, 5
    seq = synth.make_sequence(rows, cols, n, index=61, step_rot=0.004, step_trans=0.03)
    over = dict(descriptor=desc)
    if desc == "latch":
        over["levels"] = 4
    hip = bpvo_amd.load()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    runs = {}
    kw = None
    for name, bind, setup in (("hip fast", hip, None), ("hip reference-order", hip, lambda c: c.set_option("reference_reduction", 1)),
                              ("oracle serial f32", orc, None), ("oracle 8 chunks", orc, lambda c: c.call("set_num_threads", 8)),
                              ("oracle f64", orc, lambda c: c.call("set_reduction", 1))):
        p, kw = rc.params_of(bind, config, **over)
        ctx = bind.create(seq["K"], seq["b"], rows, cols, p, n_frames=2, n_pairs=1)
        if setup:
            setup(ctx)
        # frame `frame` against key frame 0 from the identity: what addFrame does for frame 1 of the sequence
        ctx.frame_set_data(0, *seq["frames"][0]); ctx.frame_set_template(0)
        ctx.frame_set_data(1, *seq["frames"][frame])
        runs[name] = ctx.estimate_pose_trace(0, 0, 1, max_records=8192)
        ctx.close()
    L = len(runs["hip fast"][1])
    p_tol, f_tol, g_tol = kw["parameterTolerance"], kw["functionTolerance"], kw["gradientTolerance"]
    print(f"conf/{config}.cfg / {desc}, frame {frame} against key frame 0, {cols}x{rows}, {L} levels; tolerances p {p_tol} f {f_tol} g {g_tol}, maxIterations {kw['maxIterations']}")
    for name, (T, st, tr) in runs.items():
        print(f"  {name:22s} iterations per level {[s['numIterations'] for s in st]} status {[STATUS.get(s['status'], s['status']) for s in st]}")
    ref = runs["oracle serial f32"]
    print("  final pose against the serial oracle:", {k: "%.2e rad / %.2e m" % pose_error(v[0], ref[0]) for k, v in runs.items() if k != "oracle serial f32"})
    assert all(np.array_equal(runs["hip reference-order"][i], ref[i]) if i != 1 else True for i in (0, 2)), "reference-order mode differs from the serial oracle"
    print("  hip reference-order == oracle serial f32: every record and the pose, bit for bit")
    for lvl in range(L - 1, -1, -1):
        print(f"level {lvl}:")
        per = {k: v[2][v[2][:, 67] == lvl] for k, v in runs.items()}
        h, o = per["hip fast"], per["oracle serial f32"]
        mx = max(len(h), len(o))
        for k in range(mx):
            if 12 < k < mx - 3:
                if k == 13:
                    print("  ...")
                continue
            row = []
            for name in ("hip fast", "oracle serial f32", "oracle 8 chunks", "oracle f64"):
                t = per[name]
                if k < len(t):
                    q = tests_at(t, k, p_tol, f_tol, g_tol)
                    row.append(f"{name}: f {q['f']:.7g} |dp| {q['dp']:.3g} sigma {t[k, 59]:.6g}" + (f" -> {q['fired']}" if q["fired"] else ""))
                else:
                    row.append(f"{name}: -")
            d = "%.1e rad" % pose_error(h[k, :16].reshape(4, 4), o[k, :16].reshape(4, 4))[0] if k < len(h) and k < len(o) else "-"
            print(f"  lin {k:3d}  d(hip, oracle) {d} | " + " | ".join(row))


if __name__ == "__main__":
    main()
