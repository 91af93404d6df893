"""CPU tests of the oracle (no GPU): golden vectors, independent numpy/scipy cross-checks of every stage, and the
behavioural quirks of the reference that the restatement has to carry."""
import glob
import os

import numpy as np
import pytest
import scipy.linalg
import scipy.ndimage

from bpvo_amd import capi, synth
from util import bits_equal, make_params, pose_error, setup_pair

import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


# ---------------------------------------------------------------------------------------------------------------- golden
def run_case(binding, g):
    levels = int(g["levels"])
    p = make_params(binding, descriptor=g["descriptor"].item().decode(), loss=g["loss"].item().decode(), levels=levels)
    rows, cols = g["imgA"].shape
    ctx = binding.create(g["K"], float(g["baseline"]), rows, cols, p, n_frames=2, n_pairs=1)
    if "formulation" in g:
        ctx.set_warp_formulation(int(g["formulation"]))
    ctx.frame_set_data(0, g["imgA"], g["dispA"])
    ctx.frame_set_template(0)
    ctx.frame_set_data(1, g["imgB"], g["dispA"])
    return ctx, levels


def check_against_golden(ctx, levels, g, exact_reduction):
    import hashlib
    for l in range(levels):
        assert np.array_equal(ctx.get_image(0, l), g[f"img_l{l}"])
        shas = ";".join(hashlib.sha256(np.ascontiguousarray(ctx.get_descriptor_channel(1, l, ch)).tobytes()).hexdigest()
                        for ch in range(ctx.Cn))
        assert shas == g[f"desc_sha_l{l}"].item().decode(), f"descriptor level {l}"
        assert bits_equal(ctx.get_descriptor_channel(1, l, 0), g[f"desc0_l{l}"])
        assert bits_equal(ctx.get_saliency(0, l), g[f"saliency_l{l}"])
        assert np.array_equal(ctx.get_point_indices(0, l), g[f"inds_l{l}"])
        assert bits_equal(ctx.get_points(0, l), g[f"points_l{l}"])
        Tn, Tni = ctx.get_normalization(0, l)
        assert bits_equal(np.stack([Tn, Tni]), g[f"norm_l{l}"])
        assert bits_equal(ctx.get_pixels(0, l), g[f"pixels_l{l}"])
        assert bits_equal(ctx.get_jacobians(0, l), g[f"jac_l{l}"])
        lin = ctx.linearize(0, 0, 1, l, g["T_lin"])
        assert np.array_equal(ctx.get_valid(0), g[f"valid_l{l}"])
        assert bits_equal(ctx.get_residuals(0), g[f"resid_l{l}"])
        assert bits_equal(ctx.get_weights(0), g[f"weights_l{l}"])
        f_norm, sigma, nv = g[f"lin_scalars_l{l}"]
        assert lin["sigma"] == np.float32(sigma) and lin["num_valid"] == int(nv)
        if exact_reduction:
            assert bits_equal(lin["H"], g[f"H_l{l}"]) and bits_equal(lin["G"], g[f"G_l{l}"]) and lin["f_norm"] == np.float32(f_norm)
        else:
            scale = np.abs(g[f"H_l{l}"]).max()
            assert np.abs(lin["H"] - g[f"H_l{l}"]).max() <= 2e-4 * scale
            assert np.abs(lin["G"] - g[f"G_l{l}"]).max() <= 2e-4 * max(np.abs(g[f"G_l{l}"]).max(), 1e-3 * scale)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(orc, path):
    g = np.load(path)
    # the golden pair stores imgB's image only (B's disparity is not used by estimatePose)
    ctx, levels = run_case(orc, g)
    check_against_golden(ctx, levels, g, exact_reduction=True)
    T, stats = ctx.estimate_pose(0, 0, 1)
    assert bits_equal(T, g["T_est"])
    assert [s["numIterations"] for s in stats] == list(g["iters"])
    assert [s["status"] for s in stats] == list(g["status"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(hip, path):
    """The HIP path against the committed vectors (no oracle in the loop)."""
    from util import ROT_TOL, trans_tol
    g = np.load(path)
    ctx, levels = run_case(hip, g)
    check_against_golden(ctx, levels, g, exact_reduction=False)
    T, stats = ctx.estimate_pose(0, 0, 1)
    rot, trans = pose_error(T, g["T_est"])
    assert rot <= ROT_TOL and trans <= trans_tol(g["K"]), (rot, trans)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_in_reference_order_matches_golden_bit_for_bit(hip, path):
    """... and with the option "reference_reduction" (the reference's f32 index-order sums, kernels_gn_ref.hip) the committed vectors are
    reproduced exactly: H, G, f_norm of the linearisation, the estimated pose, numIterations and status of every level."""
    g = np.load(path)
    ctx, levels = run_case(hip, g)
    ctx.set_option("reference_reduction", 1)
    check_against_golden(ctx, levels, g, exact_reduction=True)
    T, stats = ctx.estimate_pose(0, 0, 1)
    assert bits_equal(T, g["T_est"])
    assert [s["numIterations"] for s in stats] == list(g["iters"])
    assert [s["status"] for s in stats] == list(g["status"])


def test_golden_fixtures_present():
    assert len(GOLDEN) >= 4


# ------------------------------------------------------------------------------------------- independent cross-checks
def _fn(orc, name, restype=C.c_int):
    return orc.fn(name, restype)


def test_pyrdown_against_numpy(orc):
    rng = np.random.default_rng(0)
    for rows, cols in [(47, 156), (48, 64), (33, 37), (94, 311)]:
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        dr, dc = (rows + 1) // 2, (cols + 1) // 2
        out = np.empty((dr, dc), np.uint8)
        _fn(orc, "pyrdown_u8")(img.ctypes.data_as(C.c_void_p), rows, cols, out.ctypes.data_as(C.c_void_p))
        # cv::pyrDown: [1 4 6 4 1]^2 / 256, (s + 128) >> 8, BORDER_REFLECT_101 (numpy 'reflect'), even samples
        k = np.array([1, 4, 6, 4, 1], np.int64)
        pad = np.pad(img.astype(np.int64), 2, mode="reflect")
        h = sum(k[i] * pad[:, i:i + cols] for i in range(5))
        v = sum(k[i] * h[i:i + rows, :] for i in range(5))
        ref = ((v + 128) >> 8)[::2, ::2].astype(np.uint8)
        assert ref.shape == out.shape and np.array_equal(out, ref)
    const = np.full((40, 50), 77, np.uint8)
    out = np.empty((20, 25), np.uint8)
    _fn(orc, "pyrdown_u8")(const.ctypes.data_as(C.c_void_p), 40, 50, out.ctypes.data_as(C.c_void_p))
    assert (out == 77).all()


def test_census_against_numpy(orc):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    img[10:14, 20:30] = 100      # ties exercise the >= comparison
    out = np.empty_like(img)
    _fn(orc, "census")(img.ctypes.data_as(C.c_void_p), 37, 53, C.c_float(-1.0), out.ctypes.data_as(C.c_void_p))
    ref = np.zeros_like(img)
    offs = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]   # bpvo/census.cc:45-55 bit order
    c = img[1:-1, 1:-1]
    for b, (dy, dx) in enumerate(offs):
        nb = img[1 + dy:img.shape[0] - 1 + dy, 1 + dx:img.shape[1] - 1 + dx]
        ref[1:-1, 1:-1] |= ((nb >= c).astype(np.uint8) << b)
    assert np.array_equal(out, ref)
    assert (out[0] == 0).all() and (out[-1] == 0).all() and (out[:, 0] == 0).all() and (out[:, -1] == 0).all()


def test_gaussian_against_scipy(orc):
    rng = np.random.default_rng(2)
    src = rng.random((41, 57)).astype(np.float32)
    for sigma in (0.5, 1.6):
        out = np.empty_like(src)
        _fn(orc, "gaussian5x5_f32")(src.ctypes.data_as(C.c_void_p), 41, 57, C.c_float(sigma), out.ctypes.data_as(C.c_void_p))
        x = np.arange(5) - 2.0
        k = np.exp(-0.5 * x * x / sigma ** 2)
        k /= k.sum()
        ref = scipy.ndimage.correlate1d(scipy.ndimage.correlate1d(src.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        assert np.abs(out - ref).max() < 2e-6


def test_wide_gaussians_against_scipy(orc):
    """imsmooth with sigma >= 2.5 (7 and more taps) and GradientDescriptor's automatic kernel size: the generic forms of the
    filter engine against an f64 correlation with the same kernel (f32) and an exact integer evaluation (u8 fixed point)."""
    rng = np.random.default_rng(12)
    taps = _fn(orc, "imsmooth_taps")
    auto = _fn(orc, "auto_gauss_taps_f32")
    assert [taps(C.c_float(s)) for s in (0.3, 1.75, 2.4, 2.5, 2.6, 3.5, 3.6, 15.4)] == [5, 5, 5, 7, 7, 9, 9, 31]     # std::round: half away
    assert [auto(C.c_float(s)) for s in (0.3, 0.5, 0.75, 1.0, 2.0, 3.75)] == [3, 5, 7, 9, 17, 31]                   # cvRound(8 s + 1) | 1
    src = rng.random((37, 23)).astype(np.float32) * 255.0       # narrower than the widest kernel: repeated reflection
    for ksize, sigma in ((7, 2.6), (9, 1.0), (31, 15.4)):
        out = np.empty_like(src)
        assert _fn(orc, "gaussian_f32")(src.ctypes.data_as(C.c_void_p), 37, 23, ksize, C.c_float(sigma), out.ctypes.data_as(C.c_void_p)) == 0
        x = np.arange(ksize) - (ksize - 1) / 2.0
        k = np.exp(-0.5 * x * x / sigma ** 2).astype(np.float32).astype(np.float64)
        k = (k / k.sum()).astype(np.float32).astype(np.float64)
        ref = scipy.ndimage.correlate1d(scipy.ndimage.correlate1d(src.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
        assert np.abs(out - ref).max() < 2e-4, (ksize, np.abs(out - ref).max())
        img = rng.integers(0, 256, (37, 23), dtype=np.uint8)
        o8 = np.empty_like(img)
        assert _fn(orc, "gaussian_u8")(img.ctypes.data_as(C.c_void_p), 37, 23, ksize, C.c_float(sigma), o8.ctypes.data_as(C.c_void_p)) == 0
        ki = np.rint(k * 256.0).astype(np.int64)
        rows_ = scipy.ndimage.correlate1d(img.astype(np.int64), ki, axis=1, mode="mirror")
        ref8 = np.clip((scipy.ndimage.correlate1d(rows_, ki, axis=0, mode="mirror") + (1 << 15)) >> 16, 0, 255)
        assert np.array_equal(o8, ref8.astype(np.uint8)), ksize
    out = np.empty_like(src)
    assert _fn(orc, "gaussian_f32")(src.ctypes.data_as(C.c_void_p), 37, 23, 3, C.c_float(0.3), out.ctypes.data_as(C.c_void_p)) != 0   # not restated


def test_median_rule(orc):
    rng = np.random.default_rng(3)
    f = _fn(orc, "median", C.c_float)
    for n in (3, 4, 5, 16, 17, 1000, 1001):
        d = rng.random(n).astype(np.float32)
        got = f(d.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        s = np.sort(d)
        ref = s[n // 2] if n % 2 else np.float32((np.float32(s[n // 2 - 1] + s[n // 2])) / 2.0)
        assert got == ref
    d = np.array([5.0, 1.0], np.float32)
    assert f(d.ctypes.data_as(C.c_void_p), C.c_size_t(2)) == 5.0      # n < 3 -> data[0] (bpvo/utils.h:247-248)
    assert f(d.ctypes.data_as(C.c_void_p), C.c_size_t(0)) == 0.0      # empty -> 0


def test_solver_and_twist(orc):
    rng = np.random.default_rng(4)
    for _ in range(20):
        A = rng.standard_normal((40, 6))
        H = (A.T @ A * rng.uniform(1, 1e4)).astype(np.float32)
        G = rng.standard_normal(6).astype(np.float32) * 100
        dp = np.empty(6, np.float32)
        ok = _fn(orc, "solve")(H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p))
        assert ok == 1
        ref = np.linalg.solve(H.astype(np.float64), G.astype(np.float64))
        assert np.abs(dp - ref).max() <= 1e-3 * np.abs(ref).max()
    Z = np.zeros((6, 6), np.float32)
    G = np.ones(6, np.float32)
    dp = np.empty(6, np.float32)
    assert _fn(orc, "solve")(Z.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p)) == 0
    for _ in range(10):
        p = (rng.standard_normal(6) * [0.05, 0.05, 0.05, 0.2, 0.2, 0.2]).astype(np.float32)
        T = np.empty((4, 4), np.float32)
        _fn(orc, "twist_to_matrix", None)(p.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p))
        xi = np.zeros((4, 4))
        w, v = p[:3].astype(np.float64), p[3:].astype(np.float64)
        xi[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
        xi[:3, 3] = v
        assert np.abs(T - scipy.linalg.expm(xi)).max() < 1e-6
    T = np.empty((4, 4), np.float32)
    p0 = np.array([0, 0, 0, 1, 2, 3], np.float32)        # theta <= 1e-8 branch (bpvo/math_utils.h:163-165)
    _fn(orc, "twist_to_matrix", None)(p0.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p))
    assert np.array_equal(T[:3, 3], [1, 2, 3]) and np.array_equal(T[:3, :3], np.eye(3))


# --------------------------------------------------------------------------------------------------- pipeline properties
@pytest.mark.parametrize("descriptor", ["intensity", "bitplanes"])
def test_selection_rules(orc, descriptor):
    rows, cols, levels = 96, 128, 2
    ctx, d, p = setup_pair(orc, rows, cols, descriptor=descriptor, levels=levels, minNumPixelsForNonMaximaSuppression=rows * cols)
    for l in range(levels):
        r, c = ctx.level_size(l)
        S = ctx.get_saliency(0, l)
        inds = ctx.get_point_indices(0, l)
        n = len(inds)
        assert n % 16 == 0 and n > 0                                   # Q10
        y, x = inds // c, inds % c
        b = 3
        assert y.min() >= b and y.max() < r - b - 1 and x.min() >= b and x.max() < c - b - 1      # Q9
        assert np.all(np.diff(inds) > 0)                               # row-major scan order preserved
        assert np.all(S[y, x] >= p.minSaliency)
        assert (S[0] == 0).all() and (S[-1] == 0).all() and (S[1:-1, -1] == 0).all()
        if l == 0:      # NMS active only at level 0 here: strict max over rows -1..+1 x cols -1..+2 (Q8)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1, 2):
                    if dy == 0 and dx == 0:
                        continue
                    assert np.all(S[y, x] > S[y + dy, x + dx])
        pts = ctx.get_points(0, l)
        K = np.asarray(d["K"], np.float64).copy()
        K[:2] *= 0.5 ** l                                              # Q20
        disp = d["dispA"][(y << l), (x << l)]
        Z = (d["b"] * (2 ** l)) * K[0, 0] / disp
        assert np.allclose(pts[:, 2], Z, rtol=1e-5) and np.allclose(pts[:, 3], 1.0)
        assert np.allclose(pts[:, 0], (x - K[0, 2]) * Z / K[0, 0], rtol=1e-4, atol=1e-4)
        Tn, Tni = ctx.get_normalization(0, l)
        assert np.allclose(Tn @ Tni, np.eye(4), atol=1e-5)
        q = (Tn @ pts.T).T[:, :3]
        assert np.abs(q.mean(0)).max() < 1e-3 and abs(np.linalg.norm(q, axis=1).mean() - np.sqrt(3)) < 1e-3   # Hartley


def test_saliency_store_bug_reproduced(orc):
    """Q7: for C = 8 the accumulate variant stores to the row start: columns 4..n-1 are channel 0 only."""
    rows, cols = 48, 64
    ctx, d, _ = setup_pair(orc, rows, cols, descriptor="bitplanes", levels=1)
    S = ctx.get_saliency(0, 0)
    ch0 = ctx.get_descriptor_channel(0, 0, 0)
    ch7 = ctx.get_descriptor_channel(0, 0, 7)
    g0 = np.abs(ch0[1:-1, :-2] - ch0[1:-1, 2:]) + np.abs(ch0[:-2, 1:-1] - ch0[2:, 1:-1])
    assert np.array_equal(S[1:-1, 4:cols - 1], g0[:, 3:cols - 2])
    g7 = np.abs(ch7[1:-1, :-2] - ch7[1:-1, 2:]) + np.abs(ch7[:-2, 1:-1] - ch7[2:, 1:-1])
    # columns 0..2 = S0(n-4..n-2) + g7(n-4..n-2); column 3 = 0 + g7(n-1) with the read running into the next row
    assert np.array_equal(S[1:-1, 0:3], g0[:, cols - 5:cols - 2] + g7[:, cols - 5:cols - 2])


@pytest.mark.parametrize("loss", ["huber", "tukey", "l2"])
def test_weights_and_normal_equations(orc, loss):
    ctx, d, _ = setup_pair(orc, 96, 128, descriptor="bitplanes", loss=loss, levels=2)
    T = synth.twist_to_matrix([0.002, 0.001, -0.002, 0.01, 0.01, -0.02]).astype(np.float32)
    lin = ctx.linearize(0, 0, 1, 1, T)
    r, w, v = ctx.get_residuals(0), ctx.get_weights(0), ctx.get_valid(0)
    J = ctx.get_jacobians(0, 1).reshape(-1, 6)
    n = len(v)
    vv = np.tile(v, 8).astype(bool)
    assert np.all(r[~vv] == 0)
    absr = np.abs(r[vv]).astype(np.float32)
    med = np.sort(absr)[len(absr) // 2] if len(absr) % 2 else np.float32((np.sort(absr)[len(absr) // 2 - 1] + np.sort(absr)[len(absr) // 2]) / 2.0)
    sigma = np.float32(np.float32(1.4826) * (np.float32(1.0) + np.float32(5.0) / np.float32(len(absr) - 6))) * med
    assert lin["sigma"] == sigma
    x = r.astype(np.float32) * (np.float32(1.0) / sigma)
    if loss == "l2":
        ref = np.ones_like(r)
    elif loss == "huber":
        ref = np.float32(1.345) / np.maximum(np.abs(x), np.float32(1.345))
    else:
        q = np.float32(1.0) - (x * np.float32(1.0 / 4.685)) ** 2
        ref = np.where(np.abs(x) < np.float32(4.685), q * q, 0).astype(np.float32)
    assert np.allclose(w, ref, rtol=2e-6, atol=1e-7)
    assert np.all(w[~vv] == 1.0) or loss == "l2"                           # Q12: invalid entries keep weight 1
    wv = w.astype(np.float64) * vv
    H = (J.astype(np.float64) * wv[:, None]).T @ J.astype(np.float64)
    G = J.astype(np.float64).T @ (wv * r)
    assert np.allclose(lin["H"], H, rtol=2e-4, atol=1e-4 * np.abs(H).max())
    assert np.allclose(lin["G"], G, rtol=2e-4, atol=1e-4 * np.abs(G).max())
    assert abs(lin["f_norm"] - np.sqrt(np.sum(wv * r * r))) < 1e-3 * lin["f_norm"]
    assert lin["num_valid"] == int(v.sum()) and len(r) == 8 * n


def test_valid_mask_rule(orc):
    """valid = 0 <= floor(x) < W-1 and 0 <= floor(y) < R-1 in double (photo_error.cc:344-363, Q11)."""
    ctx, d, _ = setup_pair(orc, 96, 128, descriptor="intensity", levels=1)
    T = synth.twist_to_matrix([0.01, -0.02, 0.03, 0.3, -0.2, 0.1]).astype(np.float32)
    ctx.linearize(0, 0, 1, 0, T)
    v = ctx.get_valid(0)
    P = (d["K"].astype(np.float32) @ T[:3, :]).astype(np.float64)
    X = ctx.get_points(0, 0).astype(np.float64)
    u = X @ P.T
    x, y = u[:, 0] / u[:, 2], u[:, 1] / u[:, 2]
    ref = (np.floor(x) >= 0) & (np.floor(x) < 127) & (np.floor(y) >= 0) & (np.floor(y) < 95)
    edge = (np.abs(x - np.round(x)) < 1e-9) | (np.abs(y - np.round(y)) < 1e-9)
    assert np.array_equal(v.astype(bool)[~edge], ref[~edge])
    assert 0 < v.sum() < len(v)


def test_pose_recovery_and_iteration_bookkeeping(orc):
    ctx, d, _ = setup_pair(orc, 240, 320, descriptor="intensity", loss="huber", levels=3)
    T, stats = ctx.estimate_pose(0, 0, 1)
    rot, trans = pose_error(T, d["T_gt"])
    assert rot < 1e-3 and trans < 5e-3, (rot, trans)
    assert all(0 <= s["numIterations"] <= 50 for s in stats)
    assert all(s["status"] in (capi.STATUS_PARAMETER_TOL, capi.STATUS_FUNCTION_TOL, capi.STATUS_GRADIENT_TOL,
                               capi.STATUS_MAX_ITERATIONS) for s in stats)
    # throughput mode: tolerances 0 -> exactly K+2 linearisations per level, K iterations reported (Q2)
    K = 4
    ctx2, _, _ = setup_pair(orc, 96, 128, levels=2, maxIterations=K, parameterTolerance=0.0, functionTolerance=0.0,
                            gradientTolerance=0.0)
    _, st = ctx2.estimate_pose(0, 0, 1)
    assert [s["numIterations"] for s in st] == [K, K] and ctx2.total_linearizations() == 2 * (K + 2)
    # same frame twice: the pose stays at identity (re-projection round-off only)
    ctx3, d3, _ = setup_pair(orc, 96, 128, levels=2, descriptor="intensity", loss="l2")
    ctx3.frame_set_data(1, d3["imgA"], d3["dispA"])
    T3, st3 = ctx3.estimate_pose(0, 0, 1)
    assert np.abs(T3 - np.eye(4)).max() < 1e-4


def test_visual_odometry_sequence_cpu(orc):
    rows, cols = 120, 160
    seq = synth.make_sequence(rows, cols, 6, index=7, step_rot=0.004, step_trans=0.02)
    p = make_params(orc, descriptor="intensity", loss="huber", levels=3)
    ctx = orc.create(seq["K"], seq["b"], rows, cols, p, n_frames=3, n_pairs=1)
    res = [ctx.add_frame(i, dsp) for i, dsp in seq["frames"]]
    assert res[0]["isKeyFrame"] and res[0]["keyFramingReason"] == capi.KF_FIRST_FRAME
    assert np.array_equal(res[0]["pose"], np.eye(4, dtype=np.float32))
    traj = ctx.trajectory()
    assert traj.shape == (6, 4, 4)
    for r in res:
        assert np.array_equal(r["covariance"], np.eye(6, dtype=np.float32))     # Q16
    # relative motions track the ground truth (pose = motion of the newest frame w.r.t. the previous one)
    for k in range(1, 6):
        gt = seq["poses"][k] @ np.linalg.inv(seq["poses"][k - 1])
        rot, trans = pose_error(res[k]["pose"], gt)
        assert rot < 5e-3 and trans < 5e-2, (k, rot, trans)
    assert ctx.add_frame_null() != 0


def test_f32_formulation_close_to_f64_formulation(orc):
    """projectPoints/BilinearInterp (all float, truncation) vs PhotoError (double, floor): same residuals up to rounding on
    points that are valid in both; differences only in the handling of invalid points and of x in (-1, 0)."""
    ctx, d, _ = setup_pair(orc, 96, 128, descriptor="bitplanes", levels=2)
    T = synth.twist_to_matrix([0.004, -0.003, 0.002, 0.02, -0.015, 0.03]).astype(np.float32)
    ctx.linearize(0, 0, 1, 0, T)
    v64, r64 = ctx.get_valid(0).astype(bool), ctx.get_residuals(0).reshape(8, -1)
    ctx.set_warp_formulation(1)
    ctx.linearize(0, 0, 1, 0, T)
    v32, r32 = ctx.get_valid(0).astype(bool), ctx.get_residuals(0).reshape(8, -1)
    both_valid = v64 & v32
    assert both_valid.sum() > 0.9 * len(v64)
    assert (v64 != v32).sum() <= 0.01 * len(v64)
    assert np.abs(r64[:, both_valid] - r32[:, both_valid]).max() < 2e-4
    pix = ctx.get_pixels(0, 0)
    assert np.array_equal(r32[:, ~v32], -pix[:, ~v32]) and np.all(r64[:, ~v64] == 0)


def test_disparity_space_warp_is_the_rigid_warp_reparametrised(orc):
    """DisparitySpaceWarp (bpvo/disparity_space_warp.{h,cc}) describes the same motion model in (x - cx, y - cy, d)
    coordinates: with x' = fx X / Z, d = b fx / Z its projection H p and its Jacobian rows equal RigidBodyWarp's (without
    Hartley normalisation) up to float rounding.  Independent check of the restatement: the two warps are written from
    different source files and must agree with each other."""
    kw = dict(descriptor="bitplanes", levels=2, withNormalization=0)
    ctx, d, _ = setup_pair(orc, 96, 128, **kw)
    T = synth.twist_to_matrix([0.004, -0.003, 0.002, 0.02, -0.015, 0.03]).astype(np.float32)
    ctx.set_warp_formulation(1)
    ctx.linearize(0, 0, 1, 0, T)
    v1, r1 = ctx.get_valid(0).astype(bool), ctx.get_residuals(0).reshape(8, -1)
    J1, X1 = ctx.get_jacobians(0, 0), ctx.get_points(0, 0)
    T1, _ = ctx.estimate_pose(0, 0, 1)

    ctx.set_warp_formulation(2)
    ctx.frame_set_template(0)
    K, b = np.asarray(d["K"], np.float64), float(d["b"])
    P = ctx.get_points(0, 0)
    inds = ctx.get_point_indices(0, 0)
    assert np.array_equal(P[:, 0], (inds % 128).astype(np.float32) - np.float32(K[0, 2]))
    assert np.array_equal(P[:, 1], (inds // 128).astype(np.float32) - np.float32(K[1, 2]))
    assert np.array_equal(P[:, 2], d["dispA"].ravel()[inds]) and np.all(P[:, 3] == 1.0)
    # the same 3-D points: X = x' Z / fx, Z = b fx / d
    Z = b * K[0, 0] / P[:, 2].astype(np.float64)
    assert np.allclose(P[:, 0] * Z / K[0, 0], X1[:, 0], rtol=1e-5, atol=1e-6) and np.allclose(Z, X1[:, 2], rtol=1e-5)
    ctx.linearize(0, 0, 1, 0, T)
    v2, r2 = ctx.get_valid(0).astype(bool), ctx.get_residuals(0).reshape(8, -1)
    assert (v1 != v2).sum() <= 0.01 * len(v1)
    both_valid = v1 & v2
    assert np.abs(r1[:, both_valid] - r2[:, both_valid]).max() < 1e-3
    J2 = ctx.get_jacobians(0, 0)
    scale = np.abs(J1).max(axis=0)
    assert np.all(np.abs(J1 - J2).max(axis=0) <= 1e-4 * scale), np.abs(J1 - J2).max(axis=0) / scale
    T2, st = ctx.estimate_pose(0, 0, 1)
    rot, trans = pose_error(T1, T2)
    assert rot < 2e-4 and trans < 4e-3, (rot, trans)
    # switching back restores the rigid warp
    ctx.set_warp_formulation(0)
    ctx.frame_set_template(0)
    assert np.array_equal(ctx.get_points(0, 0), X1)


# --------------------------------------------------------------------------------------- interpolation variants (8f.3)
def _np_project(ctx, d, T, level=0):
    """f64 projection of PhotoError::Impl::init with the oracle's own P = K*T[0:3] in f32."""
    K = d["K"].astype(np.float32)
    P = np.zeros((3, 4), np.float32)
    for r in range(3):
        for c in range(4):
            s = K[r, 0] * T[0, c]
            s = np.float32(s + K[r, 1] * T[1, c])
            s = np.float32(s + K[r, 2] * T[2, c])
            P[r, c] = s
    X = ctx.get_points(0, level).astype(np.float64)
    u = X @ P.astype(np.float64).T
    zi = 1.0 / u[:, 2]
    return u[:, 0] * zi, u[:, 1] * zi


def _cubic_coeffs(x):
    x = x.astype(np.float32)
    A = np.float32(-0.5)
    one = np.float32(1)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return [c0, c1, c2, c3]


def _hermite(y, mu):
    mu = mu.astype(np.float32)
    mu2 = mu * mu
    mu3 = mu * mu2
    half = lambda a: a.astype(np.float64) / 2.0
    m0 = (half(y[1] - y[0]) + half(y[2] - y[1])).astype(np.float32)
    m1 = (half(y[2] - y[1]) + half(y[3] - y[2])).astype(np.float32)
    two, three, one = np.float32(2), np.float32(3), np.float32(1)
    a0 = two * mu3 - three * mu2 + one
    a1 = mu3 - two * mu2 + mu
    a2 = mu3 - mu2
    a3 = -two * mu3 + three * mu2
    return a0 * y[1] + a1 * m0 + a2 * m1 + a3 * y[2]


@pytest.mark.parametrize("interp", ["cosine", "cubic", "cubic_hermite"])
def test_interpolation_variants_against_numpy(orc, interp):
    """PhotoError::Impl::run for kCosine / kCubic / kCubicHermite (bpvo/photo_error.cc:391-444) restated a second time
    in numpy: borders (0,1) vs (1,3), rows yi-1..yi+2, columns xi..xi+3 (the reference's Map starts at xi), f32
    coefficient arithmetic, pairwise 4-float dot products."""
    it = {"cosine": capi.INTERP_COSINE, "cubic": capi.INTERP_CUBIC, "cubic_hermite": capi.INTERP_CUBIC_HERMITE}[interp]
    rows, cols = 96, 128
    ctx, d, _ = setup_pair(orc, rows, cols, descriptor="intensity", levels=1, interp=it)
    T = synth.twist_to_matrix([0.01, -0.02, 0.03, 0.3, -0.2, 0.1]).astype(np.float32)
    ctx.linearize(0, 0, 1, 0, T)
    v = ctx.get_valid(0).astype(bool)
    r = ctx.get_residuals(0).reshape(-1)
    x, y = _np_project(ctx, d, T)
    xi, yi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    lo, hi = (0, 1) if interp == "cosine" else (1, 3)
    ref_valid = (xi >= lo) & (xi < cols - hi) & (yi >= lo) & (yi < rows - 1)
    assert np.array_equal(v, ref_valid)
    assert 0 < v.sum() < len(v)
    I1 = ctx.get_descriptor_channel(1, 0, 0)
    I0 = ctx.get_pixels(0, 0)[0]
    xf = (x - xi).astype(np.float32)[v]
    yf = (y - yi).astype(np.float32)[v]
    xv, yv = xi[v], yi[v]
    if interp == "cosine":
        def cosc(t):
            m = (1.0 - np.cos(t.astype(np.float64) * np.pi)) / 2.0
            return (1.0 - m).astype(np.float32), m.astype(np.float32)
        cx0, cx1 = cosc(xf)
        cy0, cy1 = cosc(yf)
        d1 = I1[yv, xv] * cx0 + I1[yv, xv + 1] * cx1
        d2 = I1[yv + 1, xv] * cx0 + I1[yv + 1, xv + 1] * cx1
        Iw = cy0 * d1 + cy1 * d2
        assert np.abs((Iw - I0[v]) - r[v]).max() <= 3e-7 * 255      # libm vs numpy cos may differ in the last double bit
    else:
        rowsel = [np.minimum(yv - 1 + k, rows - 1) for k in range(4)]
        taps = [[I1[rowsel[k], xv + m] for m in range(4)] for k in range(4)]
        if interp == "cubic":
            cx, cy = _cubic_coeffs(xf), _cubic_coeffs(yf)
            dot4 = lambda a, b: (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3])
            dd = [dot4(taps[k], cx) for k in range(4)]
            Iw = dot4(cy, dd)
        else:
            V = [_hermite(taps[k], xf) for k in range(4)]
            Iw = _hermite(V, yf)
        assert Iw.dtype == np.float32
        assert bits_equal((Iw - I0[v]).astype(np.float32), r[v])
    assert np.all(r[~v] == 0.0)
    # the pose is still recovered with every interpolation type (the cubic forms carry the reference's one-column
    # offset, so they are only asked to stay in the neighbourhood)
    Te, st = ctx.estimate_pose(0, 0, 1)
    rot, trans = pose_error(Te, d["T_gt"])
    assert rot < 2e-2 and trans < 0.5, (rot, trans)


@pytest.mark.parametrize("descriptor", ["intensity", "bitplanes"])
def test_linear_interpolation_against_numpy(orc, descriptor):
    """The hot path's residual, restated a second time in numpy straight from PhotoError::Impl::init / run kLinear
    (bpvo/photo_error.cc:343-389,446-449): f64 projection with the f32 P = K T, Floor, valid = 0 <= xi < cols-1 && 0 <= yi < rows-1,
    Iw = (1-yf) (I00 wx + I01 xf) + yf (I10 wx + I11 xf) in double with wx = 1 - xf, r = float(Iw - double(I0)), 0 where invalid —
    every channel, bit for bit."""
    rows, cols = 96, 128
    ctx, d, _ = setup_pair(orc, rows, cols, descriptor=descriptor, levels=1)
    T = synth.twist_to_matrix([0.01, -0.02, 0.03, 0.3, -0.2, 0.1]).astype(np.float32)
    ctx.linearize(0, 0, 1, 0, T)
    n = ctx.num_points(0, 0)
    C = ctx.Cn
    v = ctx.get_valid(0).astype(bool).reshape(-1)
    v = v.reshape(C, n) if v.size == C * n else np.tile(v[:n], (C, 1))     # per point, or replicated per channel (replicateValidFlags)
    r = ctx.get_residuals(0).reshape(C, n)
    x, y = _np_project(ctx, d, T)
    xi, yi = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
    ref_valid = (xi >= 0) & (xi < cols - 1) & (yi >= 0) & (yi < rows - 1)
    assert 0 < ref_valid.sum() < n
    xf, yf = (x - xi)[ref_valid], (y - yi)[ref_valid]
    xv, yv = xi[ref_valid], yi[ref_valid]
    wx = 1.0 - xf
    I0 = ctx.get_pixels(0, 0)
    for c in range(C):
        assert np.array_equal(v[c], ref_valid)                       # replicateValidFlags: the same mask for every channel
        I1 = ctx.get_descriptor_channel(1, 0, c).astype(np.float64)
        Iw = (1.0 - yf) * (I1[yv, xv] * wx + I1[yv, xv + 1] * xf) + yf * (I1[yv + 1, xv] * wx + I1[yv + 1, xv + 1] * xf)
        assert bits_equal((Iw - I0[c][ref_valid].astype(np.float64)).astype(np.float32), r[c][ref_valid]), c
        assert np.all(r[c][~ref_valid] == 0.0)


def test_jacobian_is_the_derivative_of_the_warp(orc):
    """An independent check of RigidBodyWarp::computeJacobian (bpvo/rigid_body_warp.cc:60-315) together with paramsToPose
    (bpvo/rigid_body_warp.h:130-138): row k of a point's Jacobian must be Ix du/dp_k + Iy dv/dp_k of the projection
    (u, v) = normHomog(K [T_inv exp(p) T] X) at p = 0, T the Hartley normalisation of the level — here by central finite
    differences in float64 on the oracle's own points, gradients and normalisation.  (The SSE code's reciprocal is the exact one
    here, Q13; 1e-3 relative covers f32 rounding of the rows.)"""
    rows, cols = 96, 128
    ctx, d, _ = setup_pair(orc, rows, cols, descriptor="intensity", levels=1)
    X = ctx.get_points(0, 0).astype(np.float64)
    J = ctx.get_jacobians(0, 0)[0].astype(np.float64)                 # [n][6]
    Tn, Tni = (m.astype(np.float64) for m in ctx.get_normalization(0, 0))
    K = np.eye(4); K[:3, :3] = d["K"].astype(np.float64)
    inds = ctx.get_point_indices(0, 0)
    I = ctx.get_descriptor_channel(0, 0, 0).astype(np.float64)
    yy, xx = inds // cols, inds % cols
    Ix = 0.5 * (I[yy, xx + 1] - I[yy, xx - 1])                       # the central differences of TemplateData::setData (CD3)
    Iy = 0.5 * (I[yy + 1, xx] - I[yy - 1, xx])

    def project(p):
        M = K @ Tni @ synth.twist_to_matrix(p).astype(np.float64) @ Tn
        u = X @ M.T
        return u[:, 0] / u[:, 2], u[:, 1] / u[:, 2]

    h = 1e-6
    assert np.abs(J).max() > 1.0
    for k in range(6):
        e = np.zeros(6); e[k] = h
        (up, vp), (um, vm) = project(e), project(-e)
        fd = Ix * (up - um) / (2 * h) + Iy * (vp - vm) / (2 * h)
        scale = np.abs(fd).max()
        assert scale > 0
        assert np.abs(fd - J[:, k]).max() <= 1e-3 * scale, (k, np.abs(fd - J[:, k]).max(), scale)


def test_points_gradients_and_normalization_against_numpy(orc):
    """Three more pieces of the template restated independently in numpy from the reference's text:
    makePoint (bpvo/rigid_body_warp.h:47-60: Z = Bf * (1.0 / d) through double, X = (x - cx) * Z * (1.0f / fx)) — bit for bit;
    the template pixels and the Jacobian's gradients (bpvo/template_data.cc:111-131: CD3 0.5f * (I[+1] - I[-1]), CD5
    (1/18) (I[-2] - 8 I[-1] + 8 I[+1] - I[+2])) through the rows the oracle builds from them; HartlyNormalization
    (bpvo/warps.cc:27-48: c = mean point, m = mean |p - c|, s = sqrt(3.0) / max(m, 1e-6f), T = [s I, -s c; 0 1]) — to 5e-5
    against an f64 evaluation (the reference adds the N points one after the other in f32; the order in which Eigen adds the four
    squares of a norm is third-party)."""
    rows, cols = 96, 128
    for grad in (capi.GRAD_CD3, capi.GRAD_CD5):
        ctx, d, _ = setup_pair(orc, rows, cols, descriptor="intensity", levels=1, gradientEstimation=grad)
        inds = ctx.get_point_indices(0, 0)
        yy, xx = inds // cols, inds % cols
        K = d["K"].astype(np.float32)
        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        disp = d["dispA"].astype(np.float32)[yy, xx]
        Bf = np.float32(np.float32(d["b"]) * fx)
        Z = (Bf.astype(np.float64) * (1.0 / disp.astype(np.float64))).astype(np.float32)
        Xp = ((xx.astype(np.float32) - cx) * Z) * np.float32(np.float32(1.0) / fx)
        Yp = ((yy.astype(np.float32) - cy) * Z) * np.float32(np.float32(1.0) / fy)
        P = ctx.get_points(0, 0)
        assert bits_equal(P[:, 0], Xp) and bits_equal(P[:, 1], Yp) and bits_equal(P[:, 2], Z) and np.all(P[:, 3] == 1.0)
        # pixels and gradients
        I = ctx.get_descriptor_channel(0, 0, 0)
        assert bits_equal(ctx.get_pixels(0, 0)[0], I[yy, xx])
        if grad == capi.GRAD_CD3:
            Ix = np.float32(0.5) * (I[yy, xx + 1] - I[yy, xx - 1])
            Iy = np.float32(0.5) * (I[yy + 1, xx] - I[yy - 1, xx])
        else:
            NN = np.float32(np.float32(1.0) / np.float32(18.0))
            e = np.float32(8.0)
            Ix = NN * (((I[yy, xx - 2] - e * I[yy, xx - 1]) + e * I[yy, xx + 1]) - I[yy, xx + 2])
            Iy = NN * (((I[yy - 2, xx] - e * I[yy - 1, xx]) + e * I[yy + 1, xx]) - I[yy + 2, xx])
        # columns 3 and 4 of a Jacobian row are (fx Ix) / (z s) and (fy Iy) / (z s) (bpvo/rigid_body_warp.cc:205-233): the gradients
        # can be read back from them
        Tn, _ = ctx.get_normalization(0, 0)
        J = ctx.get_jacobians(0, 0)[0].astype(np.float64)
        zs = Z.astype(np.float64) * float(Tn[0, 0])
        assert np.abs(J[:, 3] * zs - float(fx) * Ix.astype(np.float64)).max() <= 2e-6 * np.abs(float(fx) * Ix).max()
        assert np.abs(J[:, 4] * zs - float(fy) * Iy.astype(np.float64)).max() <= 2e-6 * np.abs(float(fy) * Iy).max()
        # Hartley normalisation
        P64 = P.astype(np.float64)
        c = P64.mean(axis=0)
        m = np.linalg.norm(P64 - c, axis=1).mean()
        sN = np.sqrt(3.0) / max(m, 1e-6)
        assert abs(float(Tn[0, 0]) - sN) <= 5e-5 * sN and Tn[0, 0] == Tn[1, 1] == Tn[2, 2] and Tn[3, 3] == 1.0
        assert np.abs(Tn[:3, 3].astype(np.float64) + sN * c[:3]).max() <= 5e-5 * np.abs(sN * c[:3]).max()
        assert np.all(Tn[3, :3] == 0.0) and np.all(Tn[:3, :3] - np.diag(np.diag(Tn[:3, :3])) == 0.0)


@pytest.mark.parametrize("ksize", [1, 3, 5, 7])
def test_laplacian_descriptor_against_scipy(orc, ksize):
    """cv::Laplacian(u8 -> f32, ksize 1 / 3) = correlation with {0,1,0,1,-4,1,0,1,0} / {2,0,2,0,-8,0,2,0,2}; ksize 5 / 7 = Sobel
    second derivatives d2/dx2 + d2/dy2, whose separable kernels are re-derived here the way cv::getSobelKernels builds them
    (binomial smoothing [1 1]^k, differences [-1 1]); BORDER_REFLECT_101 (scipy's mode='mirror'); integer-valued, so exact."""
    rows, cols = 57, 83
    d = synth.make_pair(rows, cols, 4)
    p = make_params(orc, descriptor="laplacian", levels=2, laplacianKernelSize=ksize)
    ctx = orc.create(d["K"], d["b"], rows, cols, p, n_frames=1, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    if ksize <= 3:
        K = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.float64) if ksize == 1 else np.array([[2, 0, 2], [0, -8, 0], [2, 0, 2]], np.float64)
    else:
        def sobel_1d(order):
            k = np.array([1.0])
            for _ in range(ksize - order - 1):
                k = np.convolve(k, [1.0, 1.0])
            for _ in range(order):
                k = np.convolve(k, [-1.0, 1.0])
            return k
        d2, sm = sobel_1d(2), sobel_1d(0)
        assert list(d2) == ([1, 0, -2, 0, 1] if ksize == 5 else [1, 2, -1, -4, -1, 2, 1]) and sm.sum() == 2 ** (ksize - 1)
        K = np.outer(sm, d2) + np.outer(d2, sm)           # rows = y, columns = x
    for l in range(2):
        img = ctx.get_image(0, l).astype(np.float64)
        want = scipy.ndimage.correlate(img, K, mode="mirror")
        got = ctx.get_descriptor_channel(0, l, 0)
        assert np.array_equal(got.astype(np.float64), want)


def test_gradient_descriptor_against_numpy(orc):
    """GradientDescriptor channels (bpvo/gradient_descriptor.cc:42-63, bpvo/imgproc.h:214-265): I, 0.5 * central differences with
    one-sided 0.5 * (I1 - I0) borders; saliency = channel 0's |gradient| plus the store-bug contribution of the last channel."""
    rows, cols = 45, 70
    d = synth.make_pair(rows, cols, 9)
    ctx = orc.create(d["K"], d["b"], rows, cols, make_params(orc, descriptor="gradient", levels=1), n_frames=1, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    assert ctx.Cn == 3
    I = d["imgA"].astype(np.float32)
    Ix = np.empty_like(I); Iy = np.empty_like(I)
    Ix[:, 1:-1] = np.float32(0.5) * (I[:, 2:] - I[:, :-2]); Ix[:, 0] = np.float32(0.5) * (I[:, 1] - I[:, 0]); Ix[:, -1] = np.float32(0.5) * (I[:, -1] - I[:, -2])
    Iy[1:-1] = np.float32(0.5) * (I[2:] - I[:-2]); Iy[0] = np.float32(0.5) * (I[1] - I[0]); Iy[-1] = np.float32(0.5) * (I[-1] - I[-2])
    assert bits_equal(ctx.get_descriptor_channel(0, 0, 0), I)
    assert bits_equal(ctx.get_descriptor_channel(0, 0, 1), Ix) and bits_equal(ctx.get_descriptor_channel(0, 0, 2), Iy)


def test_descriptor_fields_against_numpy(orc):
    """DescriptorFields / DescriptorFields2ndOrder (bpvo/gradient_descriptor.cc:100-160) against a numpy / scipy restatement:
    split into the positive and the (still negative) negative part, 5-tap Gaussian with REFLECT_101 borders (scipy 'mirror'),
    and the reference's repetition of the Ixx channels where Ixy was meant."""
    from scipy.ndimage import correlate1d
    rows, cols = 45, 70
    d = synth.make_pair(rows, cols, 11)
    s1, s2 = 0.75, 1.75

    def kern(s):
        x = np.arange(5) - 2.0
        k = np.exp(-0.5 / (s * s) * x * x).astype(np.float32)
        return (k * (1.0 / k.astype(np.float64).sum())).astype(np.float32)

    def smooth(a, s):
        k = kern(s).astype(np.float64)
        return correlate1d(correlate1d(a.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")

    def xg(I):
        o = np.empty_like(I); o[:, 1:-1] = 0.5 * (I[:, 2:] - I[:, :-2]); o[:, 0] = 0.5 * (I[:, 1] - I[:, 0]); o[:, -1] = 0.5 * (I[:, -1] - I[:, -2]); return o

    def yg(I):
        o = np.empty_like(I); o[1:-1] = 0.5 * (I[2:] - I[:-2]); o[0] = 0.5 * (I[1] - I[0]); o[-1] = 0.5 * (I[-1] - I[-2]); return o

    def split(a):
        return smooth(np.where(a >= 0, a, 0.0), s2), smooth(np.where(a < 0, a, 0.0), s2)

    I0 = d["imgA"].astype(np.float64)
    I = smooth(I0, s1)
    ctx = orc.create(d["K"], d["b"], rows, cols, make_params(orc, descriptor="fields1", levels=1), n_frames=1, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    assert ctx.Cn == 5
    want = [I0, *split(xg(I)), *split(yg(I))]
    for c in range(5):
        got = ctx.get_descriptor_channel(0, 0, c)
        assert np.abs(got - want[c]).max() <= 2e-4, c
    assert ctx.get_descriptor_channel(0, 0, 2).max() <= 0.0 and ctx.get_descriptor_channel(0, 0, 1).min() >= 0.0

    ctx = orc.create(d["K"], d["b"], rows, cols, make_params(orc, descriptor="fields2", levels=1), n_frames=1, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    assert ctx.Cn == 10
    Ix, Iy = xg(I), yg(I)
    Ixx, Iyy = xg(Ix), yg(Iy)
    want = [*split(Ix), *split(Ixx), *split(Ixx), *split(Iy), *split(Iyy)]
    for c in range(10):
        got = ctx.get_descriptor_channel(0, 0, c)
        assert np.abs(got - want[c]).max() <= 2e-4, c
    assert bits_equal(ctx.get_descriptor_channel(0, 0, 4), ctx.get_descriptor_channel(0, 0, 2))
    # a pose from the 10-channel descriptor: the solver path is channel-count generic
    ctx = orc.create(d["K"], d["b"], rows, cols, make_params(orc, descriptor="fields2", levels=2, loss="huber"), n_frames=2, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"]); ctx.frame_set_template(0); ctx.frame_set_data(1, d["imgB"], d["dispB"])
    T, st = ctx.estimate_pose(0, 0, 1)
    assert np.isfinite(T).all()


def test_central_difference_against_numpy(orc):
    """CentralDifferenceDescriptor (bpvo/central_difference_descriptor.cc:36-131) without smoothing against numpy: channel
    order (rows outer, columns inner, centre skipped) and clamped shifts; with smoothing against scipy within rounding."""
    from scipy.ndimage import correlate1d
    rows, cols, R = 40, 57, 2
    d = synth.make_pair(rows, cols, 12)
    p = make_params(orc, descriptor="centraldiff", levels=1, centralDifferenceRadius=R, centralDifferenceSigmaBefore=-1.0,
                    centralDifferenceSigmaAfter=-1.0)
    ctx = orc.create(d["K"], d["b"], rows, cols, p, n_frames=1, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    assert ctx.Cn == 24
    I = d["imgA"].astype(np.float32)
    ys, xs = np.mgrid[0:rows, 0:cols]
    offs = [(ox, oy) for oy in range(-R, R + 1) for ox in range(-R, R + 1) if (ox, oy) != (0, 0)]
    for c, (ox, oy) in enumerate(offs):
        want = I - I[np.clip(ys + oy, 0, rows - 1), np.clip(xs + ox, 0, cols - 1)]
        assert bits_equal(ctx.get_descriptor_channel(0, 0, c), want), c
    # default sigmas (0.75 before on the u8 image: fixed point, rounds to u8; 1.75 after in f32)
    p = make_params(orc, descriptor="centraldiff", levels=1, centralDifferenceRadius=1)
    ctx = orc.create(d["K"], d["b"], rows, cols, p, n_frames=1, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])

    def kern(s):
        x = np.arange(5) - 2.0
        k = np.exp(-0.5 / (s * s) * x * x).astype(np.float32)
        return (k * (1.0 / k.astype(np.float64).sum())).astype(np.float64)

    def smooth(a, s):
        return correlate1d(correlate1d(a.astype(np.float64), kern(s), axis=1, mode="mirror"), kern(s), axis=0, mode="mirror")

    Is = smooth(I, 0.75)                       # the reference rounds this to u8 (fixed point): within 1 grey level
    want0 = smooth(Is - Is[np.clip(ys - 1, 0, rows - 1), np.clip(xs - 1, 0, cols - 1)], 1.75)
    assert np.abs(ctx.get_descriptor_channel(0, 0, 0) - want0).max() <= 1.5


def _fuzz_regression_cases():
    import ast
    path = os.path.join(ROOT, "tests", "tools", "fuzz_regressions.txt")
    out = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        head, brace = line.split("{", 1)
        rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
        out.append((rows, cols, scene, seed, ast.literal_eval("{" + brace.split("}", 1)[0] + "}")))
    return out


def test_unnormalised_fuzz_cases_sit_on_the_solver_fallback_edge(orc):
    """tests/tools/fuzz_regressions.txt: the randomised parity cases whose final poses differ beyond the bar although every stage up to
    the weights is bit-identical.  The ones with withNormalization = 0 (conf/tsukuba_eval.cfg:8, the default configuration of apps/eval_descriptors.cc:130) share one
    mechanism, shown here on the oracle alone: PoseEstimatorData_::solve (bpvo/pose_estimator_base.h:90-111) accepts the f32 LDLT
    solution iff (H dp).isApprox(G) and otherwise solves H + 1e-3 max(diag) I in f64 — a step that is 10-100x shorter along the weak
    directions.  For these un-normalised 6x6 systems (condition numbers of 1e5 ... 1e7) that acceptance test is a knife edge: along
    the oracle's own trace there are linearisations where relative perturbations of 2e-7 of (H, G) — the size of the difference
    between two summation orders — flip the decision.  Whichever side a run lands on decides the basin it ends in; the GPU run
    (deterministic tree + f64 combine) and the serial f32 sum of the reference differ by exactly such perturbations
    (profiles/r02_fuzz_replay.txt: a linearisation-by-linearisation replay on the GPU box)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_parity as fz
    rng = np.random.default_rng(7)

    def solve(H, G):
        H = np.ascontiguousarray(H, np.float32); G = np.ascontiguousarray(G, np.float32); dp = np.zeros(6, np.float32)
        orc.fn("solve")(H.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p))
        return dp

    cases = [c for c in _fuzz_regression_cases() if c[4].get("withNormalization", 1) == 0]
    assert len(cases) >= 6
    edges = []
    for rows, cols, scene, seed, kw in cases:
        K, b, imgA, dispA, imgB, dispB, _ = fz.make_inputs(rows, cols, scene, seed)
        kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
        ctx = orc.create(K, b, rows, cols, make_params(orc, **kw2), n_frames=2, n_pairs=1)
        if kw.get("_dspace") or kw.get("_fast_warp"):
            ctx.set_warp_formulation(2 if kw.get("_dspace") else 1)
        ctx.frame_set_data(0, imgA, dispA); ctx.frame_set_data(1, imgB, dispB); ctx.frame_set_template(0)
        _, _, tr = ctx.estimate_pose_trace(0, 0, 1)
        flips = 0
        worst_cond = 0.0
        for rec in tr:
            H, G = rec[16:52].reshape(6, 6), rec[52:58]
            if not np.isfinite(H).all() or np.abs(H).max() == 0:
                continue
            worst_cond = max(worst_cond, np.linalg.cond(H.astype(np.float64)))
            n0 = np.linalg.norm(solve(H, G))
            norms = []
            for _ in range(24):
                E = 1.0 + 2e-7 * rng.standard_normal((6, 6)); E = (E + E.T) / 2
                norms.append(np.linalg.norm(solve(H * E, G * (1.0 + 2e-7 * rng.standard_normal(6)))))
            norms = np.array(norms + [n0])
            if norms.max() > 3.0 * max(norms.min(), 1e-12) and norms.max() > 1e-6:      # both kinds of step occur: f32 solution and damped fallback
                flips += 1
        edges.append((rows, cols, kw["descriptor"], len(tr), flips, worst_cond))
        ctx.close()
    print("\nfuzz regressions (withNormalization = 0): linearisations on the solver's fallback edge", edges)
    assert all(e[5] > 5e4 for e in edges), edges                       # badly conditioned, every one
    # (the GPU run of the other two leaves the oracle's path elsewhere: one of them follows the f64-accumulating oracle to 5e-8 rad,
    #  profiles/r02_fuzz_replay.txt)
    assert sum(1 for e in edges if e[4] > 0) >= len(edges) - 2, edges


# ---- LATCH (bpvo/latch_descriptor.cc), the last DenseDescriptor of the factory ---------------------------------------------------------
def _latch_table():
    import re
    txt = open(os.path.join(ROOT, "oracle", "src", "latch_table.inc")).read()
    body = txt.split("numbers only.")[1]
    return np.array([int(v) for v in re.findall(r"-?\d+", body)], dtype=np.int64)


def _latch_offsets(table, n_ints, rotation):
    """CalcuateSums' coordinates (:170-236): the table's, or rotated by the constant key-point angle of cv::KeyPoint() (-1 degree) in f32,
    truncated and clamped to the patch."""
    t = table[:n_ints].reshape(-1, 2)
    if not rotation:
        return t.copy()
    ang = np.float32(-1.0) * np.float32(3.1415926535897932384626433832795 / np.float32(180.0))
    c, s = np.cos(ang, dtype=np.float32), np.sin(ang, dtype=np.float32)
    x, y = t[:, 0].astype(np.float32), t[:, 1].astype(np.float32)
    xr = np.trunc(x * c - y * s).astype(np.int64)
    yr = np.trunc(x * s + y * c).astype(np.int64)
    return np.stack([np.clip(xr, -24, 24), np.clip(yr, -24, 24)], axis=1)


def _latch_numpy(img, nbytes, K, rotation):
    """The definition, evaluated with whole-array operations: smoothed image, per triplet the two box sums of squared differences over
    the key-point grid, bits -> bytes -> the channel planes as the reference fills them -> 5 x 5 Gaussian of sigma 1.75."""
    R, W = img.shape
    # cv::GaussianBlur(u8, 3 x 3, sigma 2): fixed-point taps round(k * 256), (sum + 2^15) >> 16
    x = np.arange(3) - 1.0
    k = np.exp(-0.5 * x * x / 4.0).astype(np.float32)
    k = (k * np.float32(1.0 / np.float64(k.astype(np.float64).sum()))).astype(np.float32)
    ki = np.rint(k.astype(np.float64) * 256.0).astype(np.int64)
    rows = scipy.ndimage.correlate1d(img.astype(np.int64), ki, axis=1, mode="mirror")
    gray = np.clip((scipy.ndimage.correlate1d(rows, ki, axis=0, mode="mirror") + (1 << 15)) >> 16, 0, 255)
    H = 24 + K
    ny, nx = R - 2 * H - 1, W - 2 * H - 1
    C = 8 * nbytes
    planes = np.zeros((C, R, W), np.float32)
    if ny > 0 and nx > 0:
        off = _latch_offsets(_latch_table(), 6 * C, rotation).reshape(C, 3, 2)        # [triplet][a, b, c][x, y]

        def patch_sum(p, q):
            acc = np.zeros((ny, nx), np.int64)
            for iy in range(-K, K + 1):
                for ix in range(-K, K + 1):
                    A = gray[H + p[1] + iy: H + p[1] + iy + ny, H + p[0] + ix: H + p[0] + ix + nx]
                    B = gray[H + q[1] + iy: H + q[1] + iy + ny, H + q[0] + ix: H + q[0] + ix + nx]
                    acc += (A - B) ** 2
            return acc
        bits = np.stack([patch_sum(off[t, 0], off[t, 1]) < patch_sum(off[t, 2], off[t, 1]) for t in range(C)])     # [triplet][ky][kx]
        buf = np.zeros((ny * nx, nbytes), np.uint8)
        for ix in range(nbytes):
            for j in range(7, -1, -1):
                buf[:, ix] |= (bits[8 * ix + (7 - j)].reshape(-1).astype(np.uint8) << j)
        flat = buf.reshape(-1)
        for c in range(nbytes):
            vals = flat[c: c + ny * nx].reshape(ny, nx)          # `_buffer.col(c).ptr()` advanced by ONE byte per key point (:1066-1078)
            for i in range(8):
                planes[8 * c + i, H: H + ny, H: H + nx] = 255.0 * ((vals >> i) & 1) - 128.0
    xx = np.arange(5) - 2.0
    g = np.exp(-0.5 * xx * xx / 1.75 ** 2)
    g /= g.sum()
    return np.stack([scipy.ndimage.correlate1d(scipy.ndimage.correlate1d(pl.astype(np.float64), g, axis=1, mode="mirror"), g, axis=0, mode="mirror")
                     for pl in planes])


def test_latch_tables_agree():
    """The two data copies of the sampling table (product, oracle) hold the same 3072 integers — and, where the reference checkout
    exists, they are `sampling_points_arr` of bpvo/latch_descriptor.cc."""
    import re
    t = _latch_table()
    assert t.shape == (3072,) and t.min() == -24 and t.max() == 24
    prod = open(os.path.join(ROOT, "bpvo_amd", "csrc", "latch_table.h")).read().split("kLatchTable[kLatchTableInts] = {")[1].split("};")[0]
    assert np.array_equal(t, np.array([int(v) for v in re.findall(r"-?\d+", prod)]))
    ref = "/root/reference/bpvo/latch_descriptor.cc"
    if os.path.exists(ref):
        txt = open(ref).read()
        a = txt.index("{", txt.index("sampling_points_arr[]"))
        assert np.array_equal(t, np.array([int(v) for v in re.findall(r"-?\d+", txt[a: txt.index("};", a)])]))


def test_latch_rotated_offsets_are_insensitive_to_the_last_bit_of_the_cosine():
    """latchRotationInvariance: the offsets are (int)(x cos - y sin) with the cosine / sine of -1 degree.  Whether the reference's `cos(angle)`
    resolves to the float or to the double overload is a property of its toolchain; every product is at least 1e-4 from an integer, three
    orders of magnitude more than the last bit of either cosine can move it."""
    t = _latch_table().reshape(-1, 2).astype(np.float64)
    ang = float(np.float32(-1.0) * np.float32(3.1415926535897932384626433832795 / np.float32(180.0)))
    for c, s in ((np.cos(ang), np.sin(ang)), (float(np.cos(np.float32(ang), dtype=np.float32)), float(np.sin(np.float32(ang), dtype=np.float32)))):
        for v in (t[:, 0] * c - t[:, 1] * s, t[:, 0] * s + t[:, 1] * c):
            assert np.abs(v - np.rint(v)).min() > 1e-4


@pytest.mark.parametrize("nbytes,K,rotation", [(1, 1, 0), (1, 3, 0), (4, 1, 0), (2, 0, 1), (4, 2, 1)])
def test_latch_descriptor_against_a_numpy_evaluation_of_the_definition(orc, nbytes, K, rotation):
    rows, cols = 97, 131
    d = synth.make_pair(rows, cols, 3)
    p = make_params(orc, descriptor="latch", levels=2, latchNumBytes=nbytes, latchHalfSsdSize=K, latchRotationInvariance=rotation)
    ctx = orc.create(d["K"], d["b"], rows, cols, p, n_frames=2, n_pairs=1)
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    assert ctx.Cn == 8 * nbytes
    want = _latch_numpy(d["imgA"], nbytes, K, rotation)
    got = np.stack([ctx.get_descriptor_channel(0, 0, c) for c in range(8 * nbytes)])
    assert np.abs(got - want).max() < 1e-3, np.abs(got - want).max()
    assert got.min() < -100 and got.max() > 50                    # both bit values occur: the planes are not trivially zero
    # level 1 (66 x 49): narrower than two borders — no key point, a zero descriptor (the loops of :136-141 never run)
    if 49 - 2 * (24 + K) - 1 <= 0:
        assert all(np.all(ctx.get_descriptor_channel(0, 1, c) == 0.0) for c in range(8 * nbytes))
    ctx.close()
