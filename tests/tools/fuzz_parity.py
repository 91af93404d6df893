#!/usr/bin/env python3
"""Randomised stage-by-stage parity: libbpvo_hip (MI355X) against the oracle over the whole parameter space of the path —
all seven descriptors on the device path, the four interpolation types, CD3 / CD5, the three loss functions, NMS on / off,
normalisation on / off, ragged image sizes, synthetic scenes and noise images.  Every stage up to the weights must be
bit-identical, the poses within the north-star bar (scaled by the conditioning for noise images, as in
tests/test_gpu_parity.py).  Run through gpurun:

    gpurun --timeout 900 -- 'python tests/tools/fuzz_parity.py --seconds 300 --seed 1 > gpurun_out/fuzz.log 2>&1'

Prints one line per failing configuration and a summary; exit code 1 if anything failed."""
import argparse
import ctypes as C
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bpvo_amd import capi, synth  # noqa: E402
from util import ROT_TOL, TRANS_TOL, bits_equal, make_params, pose_error  # noqa: E402


def trans_tol(K):
    """The bar of the randomised cases: 1e-3 m at the benchmark calibrations (fx >= 615 px); the same image-space disagreement between
    two f32 summation orders is 615 / fx times more translation at the short focal lengths this tool draws (fx down to ~100 px)."""
    return TRANS_TOL * max(1.0, 615.0 / float(K[0][0]))

DESCRIPTORS = ["intensity", "bitplanes", "gradient", "laplacian", "fields1", "fields2", "centraldiff"]
MAX_ROWS, MAX_COLS = 200, 300
BATCH_EVERY = 0      # > 0: every n-th case also runs a small batch and compares it with the pairs run one by one


def draw(rng):
    """One random configuration.  Round 5: image sizes and saliency thresholds bounded so that the coarsest level keeps points (an
    empty template level makes BOTH sides raise and compares no pose: 'estimate-error' was 7 % of the round-4 runs), 5-level problems
    (conf/kitti_eval.cfg, conf/kitti_bitplanes.cfg) on images large enough for them, and maxIterations of 50 / 100 / 400 (the values
    the reference's configurations use: types.cc:45, conf/tsukuba_eval.cfg, conf/kitti_eval.cfg)."""
    if rng.random() < 0.05:            # a five-level pyramid needs ~28 px at its top
        rows, cols, levels = int(rng.integers(448, 520)), int(rng.integers(448, 640)), 5
    else:
        rows = int(rng.integers(48, MAX_ROWS))
        cols = int(rng.integers(64, MAX_COLS))
        levels = int(rng.integers(1, 5))
        while levels > 1 and (min(rows, cols) >> (levels - 1)) < 28:
            levels -= 1
    descriptor = DESCRIPTORS[int(rng.integers(0, len(DESCRIPTORS)))]
    # (bit-planes: channel 0 alone decides the saliency, Q7, values of a blurred bit: a threshold of 1 leaves no point)
    saliency = [0.05, 0.1, 0.3] if descriptor in ("bitplanes", "latch") else [0.05, 0.1, 1.0]
    kw = dict(descriptor=descriptor, loss=["tukey", "huber", "l2"][int(rng.integers(0, 3))], levels=levels,
              gradientEstimation=int(rng.integers(0, 2)), withNormalization=int(rng.integers(0, 2)),
              interp=int(rng.choice([0, 0, 1, 2, 3])),
              minNumPixelsForNonMaximaSuppression=int(rng.choice([1, 10**9])),
              nonMaxSuppRadius=int(rng.choice([1, 1, 2])),
              minSaliency=float(rng.choice(saliency)),
              maxTestLevel=int(rng.integers(0, levels)) if rng.random() < 0.2 else 0,
              maxIterations=int(rng.choice([50, 50, 100, 400])))
    if kw["minNumPixelsForNonMaximaSuppression"] == 1 and (min(rows, cols) >> (levels - 1)) < 40:
        kw["nonMaxSuppRadius"] = 1     # (NMS everywhere with radius 2 on a 30-pixel level keeps fewer than 16 points)
    if levels == 5:                    # the reference's own threshold (types.cc:59) or none: the 30-pixel top level keeps its points
        kw["minNumPixelsForNonMaximaSuppression"] = int(rng.choice([76800, 10**9]))
    if descriptor == "bitplanes":
        kw.update(sigmaBitPlanes=float(rng.choice([-1.0, 0.5, 1.2])), sigmaPriorToCensusTransform=float(rng.choice([-1.0, 0.8])))
    elif descriptor == "laplacian":
        kw.update(laplacianKernelSize=int(rng.choice([1, 3, 5, 7])))
    elif descriptor in ("fields1", "fields2"):        # 2.6 / 3.6: imsmooth kernels of 7 / 9 taps (the generic filter forms)
        kw.update(dfSigma1=float(rng.choice([-1.0, 0.75, 1.3, 2.6])), dfSigma2=float(rng.choice([-1.0, 1.75, 0.6, 3.6])))
    elif descriptor == "centraldiff":
        kw.update(centralDifferenceRadius=int(rng.choice([1, 1, 2, 3])),
                  centralDifferenceSigmaBefore=float(rng.choice([-1.0, 0.75, 2.7])),
                  centralDifferenceSigmaAfter=float(rng.choice([-1.0, 1.75, 3.4])))
    elif descriptor == "latch":                       # (soaks with --latch only: the draw of the fixed-seed suite is unchanged)
        kw.update(latchNumBytes=int(rng.choice([1, 1, 2, 4, 8])), latchHalfSsdSize=int(rng.choice([0, 1, 1, 2, 3])),
                  latchRotationInvariance=int(rng.integers(0, 2)))
        # a key point needs 24 + K pixels on every side (bpvo/latch_descriptor.cc:135-141): a level narrower than 2 (24 + K) + 2 has none, its
        # template is empty and BOTH sides raise — parity of the error path, but no pose compared (7 - 9 % of the round-5 soaks with --latch).
        # Keep only the levels that hold a band of key points at least 20 pixels wide.
        need = 2 * (24 + kw["latchHalfSsdSize"]) + 22
        while kw["levels"] > 1 and (min(rows, cols) >> (kw["levels"] - 1)) < need:
            kw["levels"] -= 1
        if min(rows, cols) < need:                    # (too small even at full resolution: draw the image larger)
            rows, cols = max(rows, need + 8), max(cols, need + 8)
        kw["maxTestLevel"] = min(kw["maxTestLevel"], kw["levels"] - 1)
    elif descriptor == "gradient":                    # pre-smoothing with OpenCV's automatic kernel size: 5 / 9 taps
        kw.update(sigmaPriorToCensusTransform=float(rng.choice([-1.0, -1.0, 0.5, 1.0])))
    scene = int(rng.integers(0, 3))       # 0: synthetic plane pair, 1: tiled texture + shift, 2: same with noise disparity
    # library switches outside AlgorithmParameters: the all-f32 projectPoints formulation (kLinear only) and the fused
    # residual + reduction path for frozen scales (environment variable read by bpvo_hip_create; bit-identical by design)
    kw["_fast_warp"] = bool(kw["interp"] == 0 and rng.random() < 0.15)
    kw["_dspace"] = bool(kw["interp"] == 0 and rng.random() < 0.12)     # DisparitySpaceWarp as the warp (formulation 2)
    kw["_fuse_frozen"] = bool(rng.random() < 0.3)
    return rows, cols, kw, scene, int(rng.integers(0, 1 << 30))


def is_unnormalised(kw):
    """withNormalization = 0 (conf/tsukuba_eval.cfg:8, the default configuration of apps/eval_descriptors.cc:130) or the
    DisparitySpaceWarp formulation, whose setNormalization is a no-op (bpvo/disparity_space_warp.h:87-90)."""
    return (not kw.get("withNormalization", 1)) or bool(kw.get("_dspace"))


def make_inputs(rows, cols, scene, seed):
    rng = np.random.default_rng(seed)
    if scene == 0:
        d = synth.make_pair(rows, cols, seed % 5000)
        return d["K"], float(d["b"]), d["imgA"], d["dispA"], d["imgB"], d["dispB"], 1.0
    base = synth.make_pair(160, 256, seed % 97)
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = base["imgA"][yy % 160, xx % 256].copy()
    img[rng.random((rows, cols)) < 0.02] = 255
    img2 = np.roll(img, 1, axis=1)
    if scene == 1:
        disp = (1.0 + 0.05 * xx + 0.02 * yy).astype(np.float32)
    else:
        disp = rng.uniform(-1.0, 40.0, (rows, cols)).astype(np.float32)
    K = np.array([[200.0, 0, cols / 2.0], [0, 200.0, rows / 2.0], [0, 0, 1]], np.float32)
    return K, 0.2, img, disp, img2, disp, 20.0


def at_stopped_pose(Th, trace, k, rot_tol, tr_tol):
    """Is Th the pose the traced run would have returned had it stopped at its linearisation k?  A converged run applies the step of
    that linearisation TWICE (Q1, bpvo/pose_estimator_base.h:373-393: the update after testConvergence and the one of the loop):
    T_k P P with P = T_k^-1 T_(k+1), the pose of the next record being T_k P."""
    if k + 1 >= len(trace):
        return False
    Tk, Tk1 = trace[k, :16].reshape(4, 4).astype(np.float64), trace[k + 1, :16].reshape(4, 4).astype(np.float64)
    rot, tr = pose_error(Th, Tk1 @ np.linalg.inv(Tk) @ Tk1)
    return rot <= rot_tol and tr <= tr_tol


def exact_f_norm(orc, co, level, T, sigma, loss):
    """f_norm of the linearisation at pose T with the GIVEN robust scale, its terms as LinearSystemBuilder forms them (f32) but summed
    in f64 by numpy: residuals and valid flags from the oracle at T (bit-identical to the GPU's at the same pose: the stage checks of
    check_case), weights from the oracle's ComputeWeights."""
    import ctypes as C
    co.linearize(0, 0, 1, level, T)
    r, v = co.get_residuals(0), co.get_valid(0)
    vv = np.ascontiguousarray(np.tile(v, r.size // v.size) if v.size != r.size else v).astype(np.uint16)
    w = np.empty_like(r)
    orc.fn("compute_weights")(int(loss), r.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p), C.c_size_t(r.size), C.c_float(sigma),
                              w.ctypes.data_as(C.c_void_p))
    wi = (w * vv.astype(np.float32)).astype(np.float32)
    terms = ((wi * r).astype(np.float32) * r).astype(np.float32)
    return float(np.sqrt(np.sum(terms.astype(np.float64))))


def solver_acceptance_flip(orc, ch, co, K, slack):
    """True if the two Gauss-Newton traces part at ONE linearisation whose steps differ by a factor although the iterates agreed until
    then, and the oracle's solver (PoseEstimatorData_::solve restated, bpvo_orc_solve) reproduces the GPU's step from the GPU's (H, G)."""
    _, _, trh = ch.estimate_pose_trace(0, 0, 1)
    _, _, tro = co.estimate_pose_trace(0, 0, 1)
    return acceptance_flip_in_traces(orc, trh, tro, K, slack)


def acceptance_flip_in_traces(orc, trh, tro, K, slack):
    """The rule on two traces (records of BPVO_HIP_TRACE_FLOATS floats: pose, H, G, f, sigma, valid, dp, level) — the GPU's and the oracle's."""
    import ctypes as C

    def orc_solve(rec):
        H, G = np.ascontiguousarray(rec[16:52], np.float32), np.ascontiguousarray(rec[52:58], np.float32)
        dp = np.zeros(6, np.float32)
        ok = orc.fn("solve")(H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p))
        return ok, dp

    for lvl in sorted(set(trh[:, 67].astype(int)) & set(tro[:, 67].astype(int)), reverse=True):
        h, o = trh[trh[:, 67] == lvl], tro[tro[:, 67] == lvl]
        for k in range(min(len(h), len(o))):
            r_, t_ = pose_error(h[k, :16].reshape(4, 4), o[k, :16].reshape(4, 4))
            if r_ > slack * ROT_TOL or t_ > slack * trans_tol(K):
                return False                      # the runs had left the bar before any step differed by a factor: not this rule's case
            dh, do = h[k, 61:67].astype(np.float64), o[k, 61:67].astype(np.float64)
            nh, no = np.linalg.norm(dh), np.linalg.norm(do)
            if max(nh, no) < 5.0 * min(nh, no) or max(nh, no) < 1e-5:
                continue                          # the same step, to rounding (or both at the noise floor): go on
            ok_h, dp_h = orc_solve(h[k])
            ok_o, dp_o = orc_solve(o[k])
            same_h = ok_h and np.linalg.norm(dp_h - dh) <= 1e-4 * nh
            same_o = ok_o and np.linalg.norm(dp_o - do) <= 1e-4 * no
            return bool(same_h and same_o)
        # (a level may end at different iterations on the two sides — a repeated f_norm at the noise floor — with the iterates still
        # together: the first record of the next level says whether they were)
    return False


def check(hip, orc, rows, cols, kw, scene, seed):
    """One case; the two contexts are closed whatever happens (a context left alive by a failed assertion keeps later batches of the
    same process off the estimation lanes and the team kernel)."""
    ctxs = []
    try:
        return check_case(hip, orc, rows, cols, kw, scene, seed, ctxs)
    finally:
        for ctx in ctxs:
            ctx.close()


def check_case(hip, orc, rows, cols, kw, scene, seed, ctxs):
    K, b, imgA, dispA, imgB, dispB, slack = make_inputs(rows, cols, scene, seed)
    kw = dict(kw)
    fast_warp, fuse = kw.pop("_fast_warp", False), kw.pop("_fuse_frozen", False)
    formulation = 2 if kw.pop("_dspace", False) else (1 if fast_warp else 0)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0")
    for bind in (hip, orc):
        ctx = bind.create(K, b, rows, cols, make_params(bind, **kw), n_frames=2, n_pairs=1)
        if formulation:
            ctx.set_warp_formulation(formulation)
        ctx.frame_set_data(0, imgA, dispA)
        ctx.frame_set_data(1, imgB, dispB)
        ctxs.append(ctx)
    ch, co = ctxs
    try:
        ch.frame_set_template(0)
        hip_err = None
    except capi.BpvoError as e:
        hip_err = str(e)
    try:
        co.frame_set_template(0)
        orc_err = None
    except capi.BpvoError as e:
        orc_err = str(e)
    assert (hip_err is None) == (orc_err is None), ("set_template error behaviour", hip_err, orc_err)
    if hip_err is not None:
        return "template-error"
    levels = kw["levels"]
    first = kw.get("maxTestLevel", 0)
    rng = np.random.default_rng(seed ^ 0x5eed)
    for l in range(first, levels):
        assert np.array_equal(ch.get_image(0, l), co.get_image(0, l)), ("image", l)
        for c in range(ch.Cn):
            assert bits_equal(ch.get_descriptor_channel(1, l, c), co.get_descriptor_channel(1, l, c)), ("descriptor", l, c)
        assert bits_equal(ch.get_saliency(0, l), co.get_saliency(0, l)), ("saliency", l)
        assert np.array_equal(ch.get_point_indices(0, l), co.get_point_indices(0, l)), ("indices", l)
        assert bits_equal(ch.get_points(0, l), co.get_points(0, l)), ("points", l)
        assert bits_equal(ch.get_pixels(0, l), co.get_pixels(0, l)), ("pixels", l)
        assert bits_equal(ch.get_jacobians(0, l), co.get_jacobians(0, l)), ("jacobians", l)
        if ch.num_points(0, l) == 0:
            continue
        w = rng.uniform(-0.004, 0.004, 3)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] += np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float32)
        T[:3, 3] = rng.uniform(-0.02, 0.02, 3)
        a, b2 = ch.linearize(0, 0, 1, l, T), co.linearize(0, 0, 1, l, T)
        assert np.array_equal(ch.get_valid(0), co.get_valid(0)), ("valid", l)
        assert bits_equal(ch.get_residuals(0), co.get_residuals(0)), ("residuals", l)
        assert a["sigma"] == b2["sigma"], ("sigma", l, a["sigma"], b2["sigma"])
        assert bits_equal(ch.get_weights(0), co.get_weights(0)), ("weights", l)
    try:
        Th, sh = ch.estimate_pose(0, 0, 1)
        hip_err = None
    except capi.BpvoError as e:
        hip_err = str(e)
    try:
        To, so = co.estimate_pose(0, 0, 1)
        orc_err = None
    except capi.BpvoError as e:
        orc_err = str(e)
    assert (hip_err is None) == (orc_err is None), ("estimate_pose error behaviour", hip_err, orc_err)
    if hip_err is not None:
        return "estimate-error"
    if not (np.isfinite(Th).all() and np.isfinite(To).all()):
        assert np.isfinite(Th).all() == np.isfinite(To).all(), "finite-ness differs"
        return "non-finite"
    rot, trans = pose_error(Th, To)
    if rot <= slack * ROT_TOL and trans <= slack * trans_tol(K):
        return "ok"
    # Beyond the bar although every stage agreed bit for bit: is the problem itself unstable?  Ask the oracle alone: its
    # other summation order (range-split reduction over 8 threads = the reference's TBB build) and a start
    # 1e-6 rad / 1e-6 m away from the identity.  When CPU-vs-CPU differs as much as GPU-vs-CPU, the final pose is decided by
    # rounding noise (tiny images, iteration limit reached), not by the implementation.
    co.call("set_num_threads", 8)
    To8, so8 = co.estimate_pose(0, 0, 1)
    co.call("set_num_threads", 1)
    Tp = np.eye(4, dtype=np.float32)
    Tp[0, 1], Tp[1, 0], Tp[2, 3] = -1e-6, 1e-6, 1e-6
    Top, sop = co.estimate_pose(0, 0, 1, Tp)
    rot8, trans8 = pose_error(To, To8)
    rotp, transp = pose_error(To, Top)
    rot8, trans8 = max(rot8, rotp), max(trans8, transp)
    if rot <= 8.0 * rot8 + slack * ROT_TOL and trans <= 8.0 * trans8 + slack * trans_tol(K):
        return "unstable-problem"
    # UN-NORMALISED problems (withNormalization = 0: conf/tsukuba_eval.cfg:8, the default of apps/eval_descriptors.cc:130; or the
    # DisparitySpaceWarp, whose setNormalization is a no-op): 6x6 systems with condition numbers of 1e5 .. 1e7, for which
    # PoseEstimatorData_::solve's acceptance of the f32 LDLT solution (else a damped f64 solve: a step 10-100x shorter along the weak
    # directions, bpvo/pose_estimator_base.h:90-111) flips on differences of 2e-7 of (H, G) — the size of the difference between two
    # f32 summation orders.  Ask the ORACLE how far its own final pose moves under exactly such differences: its 2-, 4- and 8-chunk
    # reductions (the reference's TBB build), its f64 accumulation, and eight replays with every linearisation's (H, G) multiplied by
    # 1 + 2e-7 N(0, 1) (bpvo_orc_set_perturbation).  The case is the problem's, not the implementation's, only if (a) those runs
    # THEMSELVES leave the bar and (b) the GPU's pose lies within 3x their spread, or is a fixed point of the oracle: restarted
    # there, the oracle stays within the bar of it.
    if is_unnormalised(dict(kw, _dspace=(formulation == 2))):
        spread_rot, spread_tr = rot8, trans8
        variants = [("threads", t) for t in (2, 4)] + [("f64", 1)] + [("perturb", s_) for s_ in range(8)]
        for kind, v in variants:
            if kind == "threads":
                co.call("set_num_threads", v)
            elif kind == "f64":
                co.call("set_reduction", 1)
            else:
                co.call("set_perturbation", v, C.c_double(2e-7))
            try:
                Tv, _ = co.estimate_pose(0, 0, 1)
            except capi.BpvoError:
                Tv = None
            finally:
                co.call("set_num_threads", 1)
                co.call("set_reduction", 0)
                co.call("set_perturbation", 0, C.c_double(0.0))
            if Tv is None or not np.isfinite(Tv).all():
                spread_rot, spread_tr = np.inf, np.inf      # (a perturbed replay that fails outright: as unstable as it gets)
                continue
            rv, tv = pose_error(To, Tv)
            spread_rot, spread_tr = max(spread_rot, rv), max(spread_tr, tv)
        leaves = spread_rot > slack * ROT_TOL or spread_tr > slack * trans_tol(K)
        within = rot <= 3.0 * spread_rot + slack * ROT_TOL and trans <= 3.0 * spread_tr + slack * trans_tol(K)
        To_r, _ = co.estimate_pose(0, 0, 1, Th)
        rot_r, trans_r = pose_error(Th, To_r)
        fixed = rot_r <= slack * ROT_TOL and trans_r <= slack * trans_tol(K)
        if leaves and (within or fixed):
            return "solver-fallback-edge"
        # ... or the flip itself, caught in the act.  The perturbed replays above find it when it is likely; a flip that only the GPU's
        # rounding draws (1 case in 3 157: profiles/r05_fuzz_soak.txt) leaves them inside the bar.  Then the two traces show it: up to
        # one linearisation k the iterates are the oracle's (inside the bar, every one of them), there the two steps differ by a factor
        # (the f32 LDLT solution accepted on one side, the damped f64 solve — 5 - 100x shorter — taken on the other), and the ORACLE'S
        # OWN SOLVER, handed the GPU's (H, G) of that linearisation, returns the GPU's step: the GPU's solve is the reference's, on
        # sums that differ from the oracle's by rounding.  From there on the two runs are different trajectories of the same algorithm;
        # the GPU's end point must still be one the oracle stays near when restarted there.
        out = solver_acceptance_flip(orc, ch, co, K, slack)
        if out:
            To_r2, _ = co.estimate_pose(0, 0, 1, Th)
            rot_n, trans_n = pose_error(Th, To_r2)
            if rot_n <= 4.0 * slack * ROT_TOL and trans_n <= 4.0 * slack * trans_tol(K):
                return "solver-acceptance-flip"
    # Last resort: both sides wander at the f32 noise floor of G (iteration limit or a repeated f_norm ends the level), on a
    # flat minimum.  Then the GPU's pose must be as good a minimum for the oracle as its own: restarted there, the oracle
    # stays within the bar of it and ends with the same weighted error.
    To2, so2 = co.estimate_pose(0, 0, 1, Th)
    rot2, trans2 = pose_error(Th, To2)
    e_own, e_at = so[first]["finalError"], so2[first]["finalError"]
    near = rot2 <= 4.0 * slack * ROT_TOL and trans2 <= 4.0 * slack * trans_tol(K)
    if near and sh[first]["status"] == capi.STATUS_MAX_ITERATIONS and so[first]["status"] == capi.STATUS_MAX_ITERATIONS:
        # neither side converged on the finest level within maxIterations: the pose after 50 steps of a still-moving
        # iteration depends on every rounding on the way; the weighted errors of two such stopping points need not agree
        return "iteration-limit"
    tol_stop = (capi.STATUS_PARAMETER_TOL, capi.STATUS_FUNCTION_TOL, capi.STATUS_GRADIENT_TOL)
    if near and sh[first]["status"] in tol_stop and so2[first]["status"] in tol_stop and so2[first]["numIterations"] <= 3:
        # the GPU stopped on one of testConvergence's tolerance tests (typically |f - f_prev| < 1e-6 on two f32 sums that
        # happen to repeat) where the oracle's own sums did not trip it; handed the GPU's pose, the oracle's tests fire
        # there at once as well: both are stopping points of the reference's rule on a flat, slowly converging problem
        return "stops-where-the-oracle-would"
    # A tolerance stop at the f32 noise floor of f_norm.  testConvergence's `|f - f_prev| < functionTolerance` (1e-6 by default) fires
    # only when two consecutive f32 values of f_norm are IDENTICAL.  Where the true change of f between two iterates is smaller than
    # the rounding error of the sum — at a turning point of a non-monotone f, or on a plateau — whether the two values coincide is
    # decided by that rounding: the GPU (tree + f64 block combine, within 4e-6 of the exact sum: test_linearize_parity) and the
    # oracle (serial f32, within 2e-4) draw differently.  Accepted when the oracle's own f64-accumulated trace shows, at the very
    # iteration the GPU stopped on, a change of f below the GPU sum's error bound: nothing but rounding decided the test.
    if sh[first]["status"] == capi.STATUS_FUNCTION_TOL and all(sh[l]["numIterations"] == so[l]["numIterations"] for l in range(first + 1, levels)):
        co.call("set_reduction", 1)
        _, _, tr64 = co.estimate_pose_trace(0, 0, 1)
        co.call("set_reduction", 0)
        t0 = tr64[tr64[:, 67] == first]
        k = sh[first]["numIterations"]          # the level's linearisations 0 .. k: the stop compared f_k with f_(k-1)
        if 1 <= k < len(t0) and abs(t0[k, 58] - t0[k - 1, 58]) <= 4e-6 * t0[k, 58]:
            # ... and the GPU's pose is what the oracle's run gives when stopped at that iterate
            if at_stopped_pose(Th, t0, k, slack * ROT_TOL, slack * trans_tol(K)):
                return "function-tol-at-the-noise-floor"
    # A GENUINE FunctionTol stop on the GPU's own sums, in an iteration that wanders: f_norm as a function of the pose has steps (a point
    # crossing the image border, 1e-7 rad apart, changes it by ~1e-3 here) and the iteration walks around its minimum with changes of that
    # size until maxIterations; two consecutive f32 values coincide once in a few hundred such steps, for the GPU's sums at one iterate,
    # for the reference's at another (or never within 50).  Accepted when (a) up to the stop the GPU's iterates ARE the oracle's (1e-6 rad /
    # 1e-5 m apart at most), (b) the stop is the reference's rule applied to correct values: at the GPU's own last two iterates, with its
    # own robust scales, the exact (f64) sums over the oracle's residuals differ by less than the error bound of the GPU's sums, and
    # (c) the GPU's pose is the oracle's iterate of that moment (Q1: T_k updated twice with dp_k).
    if sh[first]["status"] == capi.STATUS_FUNCTION_TOL and all(sh[l]["numIterations"] == so[l]["numIterations"] for l in range(first + 1, levels)):
        _, _, trh = ch.estimate_pose_trace(0, 0, 1)
        _, _, tro = co.estimate_pose_trace(0, 0, 1)
        h0, o0 = trh[trh[:, 67] == first], tro[tro[:, 67] == first]
        k = len(h0) - 1
        if k >= 1 and len(o0) > k and abs(h0[k, 58] - h0[k - 1, 58]) < kw.get("functionTolerance", 1e-6):
            together = all(r_ <= 1e-6 and t_ <= 1e-2 * trans_tol(K) for r_, t_ in
                           (pose_error(h0[i, :16].reshape(4, 4), o0[i, :16].reshape(4, 4)) for i in range(k + 1)))
            fe = [exact_f_norm(orc, co, first, h0[i, :16].reshape(4, 4), float(h0[i, 59]), make_params(orc, **kw).lossFunction) for i in (k - 1, k)]
            genuine = abs(fe[1] - fe[0]) <= 2.0 * 4e-6 * fe[1] and all(abs(fe[i] - h0[k - 1 + i, 58]) <= 4e-6 * fe[i] for i in (0, 1))
            at_iterate = at_stopped_pose(Th, o0, k, slack * ROT_TOL, slack * trans_tol(K))
            if together and genuine and at_iterate:
                return "genuine-function-tol-stop"
    # A GENUINE freeze of the robust scale on one side only (Q6: AutoScaleEstimator stops re-estimating for the rest of the level once
    # |sigma - sigma_prev| <= 1e-6, bpvo/mestimator.cc:467-490).  In a slowly converging iteration sigma moves by ~1e-5 per step and two
    # consecutive estimates come that close once in a while — for the GPU's iterates at one step, for the reference's (1e-7 rad away, the
    # medians 1e-6 apart) at another or never; from there on the two sides minimise different objectives.  Accepted when (a) up to that step
    # the GPU's iterates ARE the oracle's, (b) the decision is the reference's rule on exact values: the robust scales of the GPU's last two
    # estimates equal, bit for bit, the oracle's fresh estimates at the GPU's own poses (an exact median of bit-identical residuals), and
    # the rule fires on them exactly as the side that froze says.
    if True:
        _, _, trh = ch.estimate_pose_trace(0, 0, 1)
        _, _, tro = co.estimate_pose_trace(0, 0, 1)
        h0, o0 = trh[trh[:, 67] == first], tro[tro[:, 67] == first]

        def freeze_step(t):      # first k with |sigma_k - sigma_(k-1)| <= 1e-6: sigma_k stays for the rest of the level
            for k in range(1, len(t)):
                if abs(np.float32(t[k, 59]) - np.float32(t[k - 1, 59])) <= np.float32(1e-6):
                    return k
            return None

        kh, ko = freeze_step(h0), freeze_step(o0)
        if kw.get("loss") != "l2" and kh != ko and (kh is not None or ko is not None):
            k = min(v for v in (kh, ko) if v is not None)
            if k < min(len(h0), len(o0)):
                together = all(r_ <= 1e-6 and t_ <= 1e-2 * trans_tol(K) for r_, t_ in
                               (pose_error(h0[i, :16].reshape(4, 4), o0[i, :16].reshape(4, 4)) for i in range(k + 1)))
                fresh = [np.float32(co.linearize(0, 0, 1, first, h0[i, :16].reshape(4, 4), reset_scale=True)["sigma"]) for i in (k - 1, k)]
                exact = all(fresh[i] == np.float32(h0[k - 1 + i, 59]) for i in (0, 1))
                fires = abs(fresh[1] - fresh[0]) <= np.float32(1e-6)
                if together and exact and fires == (kh == k):
                    return "genuine-scale-freeze"
    assert near and abs(e_at - e_own) <= 2e-4 * abs(e_own), (
        "pose", rot, trans, "cpu-vs-cpu", rot8, trans8, "oracle restarted at the GPU pose", rot2, trans2, e_own, e_at,
        [s["status"] for s in sh], [s["status"] for s in so], [(s["status"], s["numIterations"]) for s in so2])
    return "noise-floor-minimum"


REFERENCE_ORDER_OUTCOMES = {"ok-bit-exact", "template-error", "estimate-error"}


def check_reference_order(hip, orc, rows, cols, kw, scene, seed):
    """One case with the library's validation mode "reference_reduction" (H, G, f summed in the reference's f32 index order,
    bpvo_amd/csrc/kernels_gn_ref.hip): NO rule, NO tolerance — every linearisation's record (pose, H, G, f_norm, robust scale, valid count,
    step), the final pose and every level's numIterations / status / finalError / firstOrderOptimality must be the oracle's bit for bit
    (NaNs compared as bits too), or both sides must raise.  Returns one of REFERENCE_ORDER_OUTCOMES; raises AssertionError naming the
    first linearisation and field that differs."""
    ctxs = []
    try:
        K, b, imgA, dispA, imgB, dispB, _ = make_inputs(rows, cols, scene, seed)
        kw = dict(kw)
        fast_warp = kw.pop("_fast_warp", False)
        kw.pop("_fuse_frozen", None)
        formulation = 2 if kw.pop("_dspace", False) else (1 if fast_warp else 0)
        os.environ.pop("BPVO_HIP_OPTIONS", None)
        for bind in (hip, orc):
            ctx = bind.create(K, b, rows, cols, make_params(bind, **kw), n_frames=2, n_pairs=1)
            ctxs.append(ctx)
            if formulation:
                ctx.set_warp_formulation(formulation)
            ctx.frame_set_data(0, imgA, dispA)
            ctx.frame_set_data(1, imgB, dispB)
        ch, co = ctxs
        ch.set_option("reference_reduction", 1)
        errs = []
        for ctx in (ch, co):
            try:
                ctx.frame_set_template(0)
                errs.append(None)
            except capi.BpvoError as e:
                errs.append(str(e))
        assert (errs[0] is None) == (errs[1] is None), ("set_template error behaviour", errs)
        if errs[0] is not None:
            return "template-error"
        runs = []
        for ctx in (ch, co):
            try:
                runs.append(ctx.estimate_pose_trace(0, 0, 1, max_records=8192))
            except capi.BpvoError as e:
                runs.append(str(e))
        assert isinstance(runs[0], str) == isinstance(runs[1], str), ("estimate_pose error behaviour", runs[0] if isinstance(runs[0], str) else None,
                                                                         runs[1] if isinstance(runs[1], str) else None)
        if isinstance(runs[0], str):
            return "estimate-error"
        (Th, sh, rh), (To, so, ro) = runs
        fields = [("T", 0, 16), ("H", 16, 52), ("G", 52, 58), ("f_norm", 58, 59), ("sigma", 59, 60), ("num_valid", 60, 61), ("dp", 61, 67), ("level", 67, 68)]
        for i in range(min(len(rh), len(ro))):
            if not bits_equal(rh[i], ro[i]):
                bad = [n for n, lo, hi in fields if not bits_equal(rh[i][lo:hi], ro[i][lo:hi])]
                raise AssertionError(("linearisation", i, "level", int(ro[i][67]), "first differing fields", bad, "hip f / sigma / valid", rh[i][58:61].tolist(),
                                      "oracle", ro[i][58:61].tolist()))
        assert len(rh) == len(ro), ("number of linearisations", len(rh), len(ro), [s_["numIterations"] for s_ in sh], [s_["numIterations"] for s_ in so])
        for l, (a, o) in enumerate(zip(sh, so)):
            assert a["numIterations"] == o["numIterations"] and a["status"] == o["status"], ("statistics", l, a, o)
            assert np.float32(a["finalError"]).tobytes() == np.float32(o["finalError"]).tobytes(), ("finalError", l, a, o)
            assert np.float32(a["firstOrderOptimality"]).tobytes() == np.float32(o["firstOrderOptimality"]).tobytes(), ("firstOrderOptimality", l, a, o)
        assert bits_equal(Th, To), ("pose", Th.tolist(), To.tolist())
        return "ok-bit-exact"
    finally:
        for ctx in ctxs:
            ctx.close()


def iteration_cells(hip, orc, rows, cols, kw, scene, seed, calibrate=False):
    """(equal, within one, total) over the levels of one case: OptimizerStatistics::numIterations AND status of the GPU run against the oracle's
    under the reference's timing tolerances (conf/perf_*.cfg: 1e-6 / 1e-4 / 1e-6), where a level ends on a tolerance test well above
    the f32 noise floor — the count is then a property of the path, not of the last bits of a sum.
    calibrate: four more numbers — (equal, within one) of the ORACLE's 8-chunk reduction (the reference's TBB build: tbb::parallel_reduce,
    bpvo/linear_system_builder.cc:91-131,233-237, restated as eight contiguous chunks summed in order) and of its f64 accumulation against the
    oracle's serial f32 run, over the same cells: the reference's OWN spread of numIterations under its own summation orders."""
    K, b, imgA, dispA, imgB, dispB, _ = make_inputs(rows, cols, scene, seed)
    kw = dict(kw, parameterTolerance=1e-6, functionTolerance=1e-4, gradientTolerance=1e-6)
    fast_warp, fuse = kw.pop("_fast_warp", False), kw.pop("_fuse_frozen", False)
    formulation = 2 if kw.pop("_dspace", False) else (1 if fast_warp else 0)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0")
    stats = []
    for bind in (hip, orc):
        ctx = bind.create(K, b, rows, cols, make_params(bind, **kw), n_frames=2, n_pairs=1)
        try:
            if formulation:
                ctx.set_warp_formulation(formulation)
            ctx.frame_set_data(0, imgA, dispA)
            ctx.frame_set_data(1, imgB, dispB)
            ctx.frame_set_template(0)
            stats.append(ctx.estimate_pose(0, 0, 1)[1])
            if calibrate and bind is orc:
                ctx.call("set_num_threads", 8)
                stats.append(ctx.estimate_pose(0, 0, 1)[1])
                ctx.call("set_num_threads", 1)
                ctx.call("set_reduction", 1)
                stats.append(ctx.estimate_pose(0, 0, 1)[1])
        except capi.BpvoError:
            stats.append(None)
        finally:
            ctx.close()
    if any(st is None for st in stats) or len(stats) != (4 if calibrate else 2):
        return (0, 0, 0, 0, 0, 0, 0) if calibrate else (0, 0, 0)
    first = kw.get("maxTestLevel", 0)

    def agree(x, y):
        pairs = list(zip(x, y))[first:]
        return (sum(1 for a, o in pairs if a["numIterations"] == o["numIterations"] and a["status"] == o["status"]),
                sum(1 for a, o in pairs if abs(a["numIterations"] - o["numIterations"]) <= 1), len(pairs))

    equal, close, total = agree(stats[0], stats[1])
    if not calibrate:
        return equal, close, total
    e8, c8, _ = agree(stats[2], stats[1])
    e64, c64, _ = agree(stats[3], stats[1])
    return equal, close, total, e8, c8, e64, c64


def check_batch(hip, rows, cols, kw, seed, options="", dirty=False, n=None):
    """bpvo_hip_batch_run of 2-5 pairs (or n) against the same pairs estimated one at a time on a fresh context: bit for bit.  `options`: more
    settings of the batch context ("team=0,lanes=1"); `dirty`: the batch context runs a batch of OTHER images first (what a buffer the run
    does not rewrite holds is then somebody else's data, not zeros)."""
    ctxs = []
    try:
        return check_batch_case(hip, rows, cols, kw, seed, ctxs, options, dirty, n)
    finally:
        os.environ.pop("BPVO_HIP_OPTIONS", None)
        for ctx in ctxs:
            ctx.close()


def check_batch_case(hip, rows, cols, kw, seed, ctxs, options="", dirty=False, n=None):
    kw = dict(kw)
    fast_warp, fuse = kw.pop("_fast_warp", False), kw.pop("_fuse_frozen", False)
    formulation = 2 if kw.pop("_dspace", False) else (1 if fast_warp else 0)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0") + ("," + options if options else "")
    n = n or 2 + seed % 4
    b = synth.make_batch(rows, cols, n, first_index=seed % 3000)
    bc = hip.create(b["K"], b["b"], rows, cols, make_params(hip, **kw), n_frames=2 * n, n_pairs=n)
    ctxs.append(bc)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0")
    if formulation:
        bc.set_warp_formulation(formulation)
    try:
        if dirty:
            o = synth.make_batch(rows, cols, n, first_index=(seed + 1234) % 3000)
            bc.batch_run(o["images"], o["disparities"])
        poses, stats = bc.batch_run(b["images"], b["disparities"])
    except capi.BpvoError:
        return "batch-error"
    sc = hip.create(b["K"], b["b"], rows, cols, make_params(hip, **kw), n_frames=2, n_pairs=1)
    ctxs.append(sc)
    if formulation:
        sc.set_warp_formulation(formulation)
    for k in range(n):
        sc.frame_set_data(0, b["images"][2 * k], b["disparities"][2 * k])
        sc.frame_set_template(0)
        sc.frame_set_data(1, b["images"][2 * k + 1], b["disparities"][2 * k + 1])
        try:
            T, st = sc.estimate_pose(0, 0, 1)
        except capi.BpvoError:
            # a template level without points: the single-pair entry point mirrors the reference's exception
            # (bpvo/template_data.cc:177), the batch skips such levels of the affected pair (bpvo_hip.hip)
            assert any(int(stats["status"][k, l]) == capi.STATUS_SOLVER_ERROR for l in range(kw["levels"])), ("batch statistics of an empty level", k)
            continue
        assert bits_equal(T, poses[k]), ("batch pose differs from the single-pair pose", k, n)
        for l in range(kw["levels"]):
            assert st[l]["numIterations"] == int(stats["numIterations"][k, l]) and st[l]["status"] == int(stats["status"][k, l]), (
                "batch statistics", k, l)
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--max-rows", type=int, default=200)
    ap.add_argument("--max-cols", type=int, default=300)
    ap.add_argument("--batch-every", type=int, default=0)
    ap.add_argument("--cells-every", type=int, default=0, help="every n-th ok case also compares the iteration counts under the timing tolerances")
    ap.add_argument("--reference-order", action="store_true", help="every case through check_reference_order: the library's reference_reduction mode, bit for bit, no rule")
    ap.add_argument("--latch", action="store_true", help="LATCH among the descriptors drawn (levels too small for a key point give an empty template on both sides)")
    args = ap.parse_args()
    global MAX_ROWS, MAX_COLS
    MAX_ROWS, MAX_COLS = args.max_rows, args.max_cols
    if args.latch:
        DESCRIPTORS.append("latch")
    import bpvo_amd
    import __graft_entry__ as ge
    hip = bpvo_amd.load()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    outcomes = {}
    by_class = {"normalised": {}, "un-normalised": {}}
    by_desc = {}
    cells = [0, 0, 0, 0, 0, 0, 0]
    fails = 0
    n = 0
    while time.time() - t0 < args.seconds and n < args.max_cases:
        rows, cols, kw, scene, seed = draw(rng)
        n += 1
        try:
            if args.reference_order:
                out = check_reference_order(hip, orc, rows, cols, kw, scene, seed)
            else:
                out = check(hip, orc, rows, cols, kw, scene, seed)
            if args.cells_every > 0 and n % args.cells_every == 0 and out == "ok":
                e, c1, t, e8, c8, e64, c64 = iteration_cells(hip, orc, rows, cols, kw, scene, seed, calibrate=True)
                cells[0] += e; cells[1] += t; cells[2] += c1; cells[3] += e8; cells[4] += c8; cells[5] += e64; cells[6] += c64
            if args.batch_every > 0 and n % args.batch_every == 0 and out == "ok":
                outb = check_batch(hip, rows, cols, kw, seed)
                outcomes["batch-" + outb] = outcomes.get("batch-" + outb, 0) + 1
        except AssertionError as e:
            out = "FAIL"
            fails += 1
            print("FAIL", rows, cols, scene, seed, kw, e.args, flush=True)
        except Exception:
            out = "EXCEPTION"
            fails += 1
            print("EXCEPTION", rows, cols, scene, seed, kw, traceback.format_exc(), flush=True)
        outcomes[out] = outcomes.get(out, 0) + 1
        cls = by_class["un-normalised" if is_unnormalised(kw) else "normalised"]
        cls[out] = cls.get(out, 0) + 1
        by_desc[kw["descriptor"]] = by_desc.get(kw["descriptor"], 0) + 1
    print("cases", n, "seconds", round(time.time() - t0, 1), "outcomes", outcomes, "per descriptor", by_desc, flush=True)
    print("by class", by_class, flush=True)
    # rules fired per 1000 cases of each class (a rule that starts to grow is seen here first)
    for cls_name, cls in by_class.items():
        tot = max(1, sum(cls.values()))
        print("  per 1000 %s cases (%d):" % (cls_name, tot), {k: round(1000.0 * v / tot, 1) for k, v in sorted(cls.items()) if k != "ok"}, flush=True)
    if cells[1]:
        print("(case, level) cells with numIterations and status equal to the oracle's under the timing tolerances: %d of %d = %.4f; numIterations within one: %d = %.4f" % (cells[0], cells[1], cells[0] / cells[1], cells[2], cells[2] / cells[1]), flush=True)
        print("  the oracle's own spread over the same cells (against its serial f32 run): 8-chunk reduction equal %d = %.4f, within one %d = %.4f; f64 accumulation equal %d = %.4f, within one %d = %.4f"
              % (cells[3], cells[3] / cells[1], cells[4], cells[4] / cells[1], cells[5], cells[5] / cells[1], cells[6], cells[6] / cells[1]), flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
