"""Stereo front-end: block matching, the reference's default StereoAlgorithm (utils/stereo_algorithm.cc:63-111 ->
OpenCV 2.4 cvFindStereoCorrespondenceBM + disp16 / 16).  OpenCV is third party and absent: PARITY UNPINNED.  What is checked:
  * CPU: the oracle's restatement (sliding-window definition evaluated per pixel) against an independent numpy evaluation of the
    published algorithm, the pre-filter against numpy, sub-pixel accuracy against the true disparity of a rendered plane;
  * GPU (-m gpu): the HIP matcher against the oracle bit for bit — window sizes, disparity ranges, minDisparity, odd sizes, batches —
    and addFrame fed by the device-resident disparity against addFrame fed the same map from the host."""
import ctypes as C

import numpy as np
import pytest

from bpvo_amd import capi, synth


def _params(cap=31, wsz=15, mind=0, ndisp=64, tex=10, uniq=15):
    return (C.c_int * 6)(cap, wsz, mind, ndisp, tex, uniq)


def orc_bm(orc, left, right, **kw):
    out = np.empty(left.shape, np.float32)
    rc = orc.lib.bpvo_orc_stereo_bm(left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), left.shape[0], left.shape[1], _params(**kw),
                                    out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def np_prefilter(img, cap):
    """prefilterXSobel of OpenCV 2.4 from its definition: [1 2 1]^T x [-1 0 1], rows reflected, clipped to +-cap, + cap."""
    I = img.astype(np.int32)
    R, W = I.shape
    up = np.vstack([I[1:2], I[:-1]])          # row y-1 (row 1 for y = 0)
    dn = np.vstack([I[1:], I[-2:-1]])         # row y+1 (row R-2 for y = R-1)
    gx = lambda A: A[:, 2:] - A[:, :-2]
    v = gx(up) + 2 * gx(I) + gx(dn)
    out = np.full((R, W), cap, np.int32)
    out[:, 1:-1] = np.clip(v, -cap, cap) + cap
    if R % 2:
        out[-1] = cap                         # the rows are walked in pairs: an unpaired last row is left at cap
    return out.astype(np.uint8)


def np_bm(L, Rm, wsz, ndisp, mind, cap, tex, uniq):
    """Block matching from the definition (SAD over the clamped window, first minimum, texture and uniqueness tests, parabola step)."""
    rows, cols = L.shape
    w2 = wsz // 2
    lofs = max(ndisp - 1 + mind, 0)
    width1 = min(cols - ndisp + 1, cols - lofs)      # the original's range overruns the row for minDisparity > 0: cut (documented choice)
    filt = np.int16((mind - 1) << 4)
    out = np.full((rows, cols), filt, np.int16)
    Rl = Rm.reshape(-1).astype(np.int32)
    Li = L.astype(np.int32)
    ys = np.clip(np.arange(-w2, rows + w2), 0, rows - 1)
    for x in range(width1):
        xcs = np.arange(x - w2, x + w2 + 1)
        lc = np.clip(lofs + xcs, 0, cols - 1)
        rc = np.clip(xcs, 0, cols - 1)
        Lw = Li[ys][:, lc]                                             # [rows + 2 w2, wsz]
        tex_col = np.abs(Lw - cap).sum(axis=1)
        sad_rows = np.empty((rows + 2 * w2, ndisp), np.int64)
        for d in range(ndisp):
            idx = np.minimum(ys[:, None] * cols + rc[None, :] + d, rows * cols - 1)
            sad_rows[:, d] = np.abs(Lw - Rl[idx]).sum(axis=1)
        cs = np.vstack([np.zeros((1, ndisp), np.int64), np.cumsum(sad_rows, axis=0)])
        sad = cs[wsz:] - cs[:-wsz]                                      # [rows, ndisp]
        ct = np.concatenate([[0], np.cumsum(tex_col)])
        tsum = ct[wsz:] - ct[:-wsz]
        for y in range(rows):
            s = sad[y]
            md = int(np.argmin(s))
            if tsum[y] < tex:
                continue
            thresh = s[md] + (s[md] * uniq // 100)
            other = np.ones(ndisp, bool)
            other[max(md - 1, 0): md + 2] = False
            if uniq > 0 and np.any(s[other] <= thresh):
                continue
            p = s[md + 1] if md + 1 < ndisp else s[ndisp - 2]
            n = s[md - 1] if md > 0 else s[1]
            dd = p + n - 2 * s[md] + abs(p - n)
            frac = int((p - n) * 256 / dd) if dd != 0 else 0           # C integer division truncates towards zero
            out[y, lofs + x] = np.int16(((ndisp - md - 1 + mind) * 256 + frac + 15) >> 4)
    return out.astype(np.float32) / 16.0


def test_prefilter_matches_numpy(orc):
    rng = np.random.default_rng(5)
    for rows, cols in ((37, 53), (40, 64), (9, 20)):
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        for cap in (31, 15, 63):
            out = np.empty_like(img)
            assert orc.lib.bpvo_orc_stereo_prefilter(img.ctypes.data_as(C.c_void_p), rows, cols, cap, out.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(out, np_prefilter(img, cap)), (rows, cols, cap)


@pytest.mark.parametrize("rows,cols,wsz,ndisp,mind", [(41, 96, 9, 32, 0), (40, 90, 15, 16, 0), (33, 80, 5, 48, 2), (30, 70, 21, 16, 1)])
def test_block_matching_oracle_matches_the_definition(orc, rows, cols, wsz, ndisp, mind):
    d = synth.make_stereo_pair(rows, cols, 2, z0=3.0)
    rng = np.random.default_rng(rows)
    left = d["left"].copy()
    right = d["right"].copy()
    right[: rows // 3] = rng.integers(0, 256, (rows // 3, cols), dtype=np.uint8)      # a textured band without a match: exercises the uniqueness test
    left[rows // 2: rows // 2 + 6, 10:40] = 128                                       # a flat patch: the texture test
    right[rows // 2: rows // 2 + 6, 10:40] = 128
    got = orc_bm(orc, left, right, wsz=wsz, ndisp=ndisp, mind=mind)
    lp, rp = np_prefilter(left, 31), np_prefilter(right, 31)
    want = np_bm(lp, rp, wsz, ndisp, mind, 31, 10, 15)
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    assert (got == mind - 1).any() and (got > mind).any()


def test_block_matching_recovers_the_disparity_of_a_plane(orc):
    rows, cols = 120, 320
    d = synth.make_stereo_pair(rows, cols, 0, z0=6.0)
    out = orc_bm(orc, d["left"], d["right"], ndisp=64)
    valid = out >= 0
    assert valid.mean() > 0.7
    err = np.abs(out[valid] - d["disp"][valid])
    assert np.median(err) < 0.06 and np.percentile(err, 95) < 0.25, (np.median(err), np.percentile(err, 95))
    assert (out[:, :63] == -1).all()                 # the first numberOfDisparities - 1 columns have no full range: FILTERED


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,wsz,ndisp,mind", [(376, 1241, 15, 128, 0), (480, 640, 15, 64, 0), (121, 163, 9, 32, 0), (120, 160, 21, 16, 2),
                                                      (97, 203, 5, 48, 0), (376, 1241, 9, 128, 0), (64, 300, 11, 256, 0)])
def test_hip_block_matching_bit_exact(hip, orc, rows, cols, wsz, ndisp, mind):
    d = synth.make_stereo_pair(rows, cols, 4, z0=8.0 if cols > 700 else 4.0)
    rng = np.random.default_rng(cols)
    left, right = d["left"].copy(), d["right"].copy()
    right[: rows // 4] = rng.integers(0, 256, (rows // 4, cols), dtype=np.uint8)
    left[rows // 2: rows // 2 + 20, 30: 30 + cols // 4] = 100
    right[rows // 2: rows // 2 + 20, 30: 30 + cols // 4] = 100
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(d["K"], d["b"], rows, cols, p, n_frames=3, n_pairs=1)
    sp = ctx.default_stereo_params(ndisp)
    sp.SADWindowSize = wsz; sp.minDisparity = mind
    got = ctx.stereo_bm(left, right, sp)
    want = orc_bm(orc, left, right, wsz=wsz, ndisp=ndisp, mind=mind)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got != want)[:8]
    assert (got == mind - 1).any() and (got > mind).mean() > 0.2 * (cols - ndisp) / cols


@pytest.mark.gpu
def test_hip_block_matching_batch_and_errors(hip, orc):
    rows, cols, n = 120, 200, 5
    pairs = [synth.make_stereo_pair(rows, cols, k, z0=4.0) for k in range(n)]
    L = np.stack([q["left"] for q in pairs]); R = np.stack([q["right"] for q in pairs])
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(pairs[0]["K"], pairs[0]["b"], rows, cols, p, n_frames=3, n_pairs=1)
    sp = ctx.default_stereo_params(32)
    got = ctx.stereo_bm(L, R, sp)
    for k in range(n):
        assert np.array_equal(got[k], orc_bm(orc, L[k], R[k], ndisp=32)), k
    for field, bad in (("numberOfDisparities", 20), ("SADWindowSize", 8), ("SADWindowSize", 3), ("preFilterCap", 0), ("preFilterCap", 64),
                       ("SADWindowSize", 23), ("minDisparity", -3), ("numberOfDisparities", 512)):
        sp2 = ctx.default_stereo_params(32)
        setattr(sp2, field, bad)
        with pytest.raises(capi.BpvoError):
            ctx.stereo_bm(L[0], R[0], sp2)
    # numberOfDisparities wider than the image: nothing to match, every pixel FILTERED (stereobm.cpp)
    sp3 = ctx.default_stereo_params(208)
    assert (ctx.stereo_bm(L[0], R[0], sp3) == -1).all()


@pytest.mark.gpu
def test_add_frame_stereo_equals_add_frame_with_the_same_disparity(hip, orc):
    """VisualOdometry::addFrame(left, StereoAlgorithm::run(left, right)): the device-resident disparity gives the poses, key-frame
    decisions and point clouds of the same map handed over from the host."""
    rows, cols, n = 240, 640, 5
    K, b = synth.calibration(rows, cols)
    rng = np.random.default_rng(3)
    T = np.eye(4)
    frames = []
    for k in range(n):
        left, _ = synth._render(K, b, rows, cols, T, 1003, 4.0, (0.1, -0.15))
        Tr = np.eye(4); Tr[0, 3] = -b
        right, _ = synth._render(K, b, rows, cols, Tr @ T, 1003, 4.0, (0.1, -0.15))
        frames.append((left, right))
        T = synth.twist_to_matrix(np.concatenate([rng.uniform(-0.004, 0.004, 3), rng.uniform(-0.02, 0.02, 3)])) @ T
    p = hip.default_params(); p.numPyramidLevels = 3; p.verbosity = capi.VERB_SILENT; p.descriptor = capi.DESC_BITPLANES
    a = hip.create(K, b, rows, cols, p, n_frames=3, n_pairs=1)
    c = hip.create(K, b, rows, cols, p, n_frames=3, n_pairs=1)
    sp = a.default_stereo_params(48)
    for left, right in frames:
        ra = a.add_frame_stereo(left, right, sp)
        disp = orc_bm(orc, left, right, ndisp=48)
        rc = c.add_frame(left, disp)
        assert np.array_equal(ra["pose"].view(np.uint32), rc["pose"].view(np.uint32))
        assert ra["isKeyFrame"] == rc["isKeyFrame"] and ra["stats"] == rc["stats"]
    assert np.array_equal(a.trajectory(), c.trajectory()) and a.vo_num_points_at_level() == c.vo_num_points_at_level() > 0


# ---- semi-global matching: the reference's in-tree SgmStereo (utils/sgm.cc), BPVO_STEREO_SGM -----------------------------------------
SGM_DEFAULT = dict(ndisp=64, cap=15, crad=2, wrad=2, p1=100, p2=1600, thr=1, factor=256.0, cw=1.0 / 6.0)


def orc_sgm(orc, left, right, **kw):
    q = dict(SGM_DEFAULT, **kw)
    out = np.empty(left.shape, np.float32)
    ip = (C.c_int * 7)(q["ndisp"], q["cap"], q["crad"], q["wrad"], q["p1"], q["p2"], q["thr"])
    dp = (C.c_double * 2)(q["factor"], q["cw"])
    rc = orc.lib.bpvo_orc_stereo_sgm(left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p), left.shape[0], left.shape[1], ip, dp,
                                     out.ctypes.data_as(C.c_void_p))
    return out if rc == 0 else None


def np_sgm(L, R, ndisp, cap, crad, wrad, p1, p2, thr, factor, cw):
    """The algorithm of utils/sgm.cc from its definitions, written independently of the oracle: vectorised numpy over rows / disparities, the
    scanline recursions as plain loops.  Returns the float disparity map."""
    H, W = L.shape
    D = ndisp
    cap = min(max(cap, 15), 127) | 1
    Li, Ri = L.astype(np.int64), R.astype(np.int64)

    def sobel(I):
        S = np.full((H, W), cap, np.int64)
        v = (I[:-2, 2:] + 2 * I[1:-1, 2:] + I[2:, 2:]) - (I[:-2, :-2] + 2 * I[1:-1, :-2] + I[2:, :-2])
        S[1:-1, 1:-1] = np.where(v > cap, 2 * cap, np.where(v < -cap, 0, v + cap))
        return S

    def census(I):
        code = np.zeros((H, W), np.int64)
        P = np.full((H + 2 * crad, W + 2 * crad), -1, np.int64)
        P[crad:crad + H, crad:crad + W] = I
        for oy in range(-crad, crad + 1):
            for ox in range(-crad, crad + 1):
                nb = P[crad + oy: crad + oy + H, crad + ox: crad + ox + W]
                code = (code << 1) + (nb >= I).astype(np.int64) * (nb >= 0)
        return code

    def halfmm(S):
        l = np.concatenate([S[:, :1], (S[:, 1:] + S[:, :-1]) // 2], axis=1)
        r = np.concatenate([(S[:, :-1] + S[:, 1:]) // 2, S[:, -1:]], axis=1)
        return np.minimum(np.minimum(l, r), S), np.maximum(np.maximum(l, r), S)

    SL, SR = sobel(Li), sobel(Ri)
    SRf = SR[:, ::-1].copy()
    SRf[:, 0] = cap; SRf[:, -1] = cap                   # the mirrored interior leaves both border columns at `cap`
    CL, CR = census(Li), census(Ri)
    lmin, lmax = halfmm(SL)
    rminf, rmaxf = halfmm(SRf)
    pc = np.zeros((H, W, D), np.int64)
    xs = np.arange(W)
    pop = np.vectorize(lambda v: bin(int(v)).count("1"))
    for d in range(D):
        dd = np.minimum(d, xs)                           # costs past d = x repeat the cost at d = x
        ri = W - 1 - xs + dd
        rc, rmn, rmx = SRf[:, ri], rminf[:, ri], rmaxf[:, ri]
        l2r = np.maximum(np.maximum(0, SL - rmx), rmn - SL)
        r2l = np.maximum(np.maximum(0, rc - lmax), lmin - rc)
        ham = pop(CL ^ CR[:, xs - dd])
        pc[:, :, d] = (np.minimum(l2r, r2l) + (ham * cw).astype(np.int64).astype(np.uint8)) % 256
    yy = np.clip(np.arange(-wrad, H + wrad), 0, H - 1)
    xx = np.clip(np.arange(-wrad, W + wrad), 0, W - 1)
    cost = np.zeros((H, W, D), np.int64)
    for oy in range(2 * wrad + 1):
        for ox in range(2 * wrad + 1):
            cost += pc[yy[oy: oy + H]][:, xx[ox: ox + W]]
    cost[H - wrad:] = 0                                  # rows the original never writes
    cost[1:, 0] = 0                                      # ... and the first column of every row but the first
    rcost = np.zeros_like(cost)
    for x in range(W):
        for d in range(D):
            dd = min(d, W - 1 - x)
            rcost[:, x, d] = cost[:, x + dd, dd]

    def sat(v):
        return np.clip(v, -32768, 32767)

    def aggregate(Cv):
        S = np.zeros((H, W, D), np.int64)
        big = 32767

        def step(prev, pmin, c):                         # prev [..., D], pmin [...], c [..., D]
            pm = (pmin + p2)[..., None]
            lm = np.concatenate([np.full(prev.shape[:-1] + (1,), big), prev[..., :-1]], axis=-1)
            lp = np.concatenate([prev[..., 1:], np.full(prev.shape[:-1] + (1,), big)], axis=-1)
            a = np.minimum(np.minimum(prev, sat(lm + p1)), np.minimum(sat(lp + p1), pm))
            a = sat(sat(a - pm) + c)
            return a, a.min(axis=-1)

        for order in (1, -1):
            xs_ = range(W) if order == 1 else range(W - 1, -1, -1)
            ys_ = range(H) if order == 1 else range(H - 1, -1, -1)
            prev, pmin = np.zeros((H, D), np.int64), np.zeros(H, np.int64)          # along the rows, all rows at once
            for x in xs_:
                prev, pmin = step(prev, pmin, Cv[:, x, :])
                S[:, x, :] = sat(S[:, x, :] + prev)
            prev, pmin = np.zeros((W, D), np.int64), np.zeros(W, np.int64)          # along the columns, all columns at once
            for y in ys_:
                prev, pmin = step(prev, pmin, Cv[y])
                S[y] = sat(S[y] + prev)
        bd = S.argmin(axis=2)                            # first minimum
        out = np.zeros((H, W), np.int64)
        for y in range(H):
            for x in range(W):
                b = int(bd[y, x])
                if 0 < b < D - 1:
                    c, l, r = float(S[y, x, b]), float(S[y, x, b - 1]), float(S[y, x, b + 1])
                    den = (c - l) if r < l else (c - r)
                    if den == 0.0:
                        out[y, x] = 0                    # INT_MIN truncated to 16 bits
                        continue
                    out[y, x] = int(b * factor + (r - l) / den / 2.0 * factor + 0.5) & 0xffff
                else:
                    out[y, x] = int(b * factor)
        # speckle filter: components of the "both non-zero and within 2 * factor" 4-neighbour relation, at most 100 pixels -> 0
        lab = -np.ones((H, W), np.int64)
        md = int(2 * factor)
        for y0 in range(H):
            for x0 in range(W):
                if out[y0, x0] == 0 or lab[y0, x0] >= 0:
                    continue
                comp, stack = [], [(y0, x0)]
                lab[y0, x0] = 1
                while stack:
                    y, x = stack.pop()
                    comp.append((y, x))
                    for y2, x2 in ((y, x + 1), (y, x - 1), (y + 1, x), (y - 1, x)):
                        if 0 <= y2 < H and 0 <= x2 < W and lab[y2, x2] < 0 and out[y2, x2] != 0 and abs(out[y, x] - out[y2, x2]) <= md:
                            lab[y2, x2] = 1
                            stack.append((y2, x2))
                if len(comp) <= 100:
                    for y, x in comp:
                        lab[y, x] = 2
        out[lab == 2] = 0
        return out

    dl, dr = aggregate(cost), aggregate(rcost)
    res = dl.copy()
    for y in range(H):
        for x in range(W):
            if res[y, x] == 0:
                continue
            ld = int(res[y, x] / factor + 0.5)
            if x - ld < 0:
                res[y, x] = 0
                continue
            rd = int(dr[y, x - ld] / factor + 0.5)
            if rd == 0 or abs(ld - rd) > thr:
                res[y, x] = 0
    return (res / factor).astype(np.float32)


@pytest.mark.parametrize("rows,cols,kw", [(40, 72, dict(ndisp=16)), (33, 61, dict(ndisp=32, crad=1, wrad=1, p1=60, p2=900)),
                                          (36, 50, dict(ndisp=16, cap=40, thr=2, cw=0.5, factor=16.0))])
def test_sgm_oracle_against_a_numpy_evaluation_of_the_definition(orc, rows, cols, kw):
    d = synth.make_stereo_pair(rows, cols, 6, z0=2.5)
    rng = np.random.default_rng(rows)
    left, right = d["left"].copy(), d["right"].copy()
    right[: rows // 4] = rng.integers(0, 256, (rows // 4, cols), dtype=np.uint8)        # a band without a match: left-right check, speckles
    left[rows // 2: rows // 2 + 5, 8:30] = 128                                           # a flat patch: ties, zero sub-pixel denominators
    right[rows // 2: rows // 2 + 5, 8:30] = 128
    got = orc_sgm(orc, left, right, **kw)
    q = dict(SGM_DEFAULT, **kw)
    want = np_sgm(left, right, q["ndisp"], q["cap"], q["crad"], q["wrad"], q["p1"], q["p2"], q["thr"], q["factor"], q["cw"])
    assert np.array_equal(got, want), (np.argwhere(got != want)[:6], got[got != want][:6], want[got != want][:6])
    assert (got == 0).any() and (got > 0).mean() > 0.3


def test_sgm_recovers_the_disparity_of_a_plane_and_rejects_bad_arguments(orc):
    rows, cols = 120, 320
    d = synth.make_stereo_pair(rows, cols, 1, z0=6.0)
    out = orc_sgm(orc, d["left"], d["right"], ndisp=32)
    valid = out > 0
    assert valid.mean() > 0.9
    err = np.abs(out[valid] - d["disp"][valid])
    assert np.median(err) < 0.15 and np.percentile(err, 95) < 0.5, (np.median(err), np.percentile(err, 95))
    # (the last windowRadius rows have no data cost in the original — utils/sgm.cc:517-526 never writes them — and get their disparities
    # from the smoothness terms alone: plausible values, not 0)
    for bad in (dict(ndisp=24), dict(crad=3), dict(p1=200, p2=100), dict(thr=-1), dict(factor=0.0)):
        assert orc_sgm(orc, d["left"], d["right"], **bad) is None, bad


def _sgm_params(ctx, **kw):
    q = dict(SGM_DEFAULT, **kw)
    sp = ctx.default_stereo_params(q["ndisp"])
    sp.algorithm = capi.STEREO_SGM
    sp.sobelCapValue, sp.censusRadius, sp.windowRadius = q["cap"], q["crad"], q["wrad"]
    sp.smoothnessPenaltySmall, sp.smoothnessPenaltyLarge, sp.consistencyThreshold = q["p1"], q["p2"], q["thr"]
    sp.disparityFactor, sp.censusWeightFactor = q["factor"], q["cw"]
    return sp


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,kw", [(376, 1241, dict(ndisp=128)), (480, 640, dict(ndisp=64)), (121, 163, dict(ndisp=32, crad=1, wrad=1, p1=60, p2=900)),
                                          (97, 203, dict(ndisp=48, cap=40, thr=2, cw=0.5)), (64, 300, dict(ndisp=256, factor=64.0)),
                                          (50, 70, dict(ndisp=16, wrad=3)),
                                          # the largest window and Sobel cap the device path admits: box sums up to 121 * 255 = 30855, just
                                          # below the int16 range in which the original's saturating additions would start to clip
                                          (90, 160, dict(ndisp=32, wrad=5, cap=127, cw=1.0 / 6.0)), (80, 140, dict(ndisp=144, wrad=4, cap=99))])
def test_hip_sgm_bit_exact(hip, orc, rows, cols, kw):
    d = synth.make_stereo_pair(rows, cols, 4, z0=8.0 if cols > 700 else 4.0)
    rng = np.random.default_rng(cols)
    left, right = d["left"].copy(), d["right"].copy()
    right[: rows // 4] = rng.integers(0, 256, (rows // 4, cols), dtype=np.uint8)
    left[rows // 2: rows // 2 + 20, 30: 30 + cols // 4] = 100
    right[rows // 2: rows // 2 + 20, 30: 30 + cols // 4] = 100
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(d["K"], d["b"], rows, cols, p, n_frames=3, n_pairs=1)
    got = ctx.stereo_bm(left, right, _sgm_params(ctx, **kw))
    want = orc_sgm(orc, left, right, **kw)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (np.argwhere(got != want)[:8], got[got != want][:8], want[got != want][:8])
    assert (got == 0).any() and (got > 0).mean() > 0.3


@pytest.mark.gpu
def test_hip_sgm_batch_errors_and_add_frame(hip, orc):
    rows, cols, n = 120, 200, 3
    pairs = [synth.make_stereo_pair(rows, cols, k, z0=4.0) for k in range(n)]
    L = np.stack([q["left"] for q in pairs]); R = np.stack([q["right"] for q in pairs])
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(pairs[0]["K"], pairs[0]["b"], rows, cols, p, n_frames=3, n_pairs=1)
    got = ctx.stereo_bm(L, R, _sgm_params(ctx, ndisp=32))
    for k in range(n):
        assert np.array_equal(got[k], orc_sgm(orc, L[k], R[k], ndisp=32)), k
    for bad in (dict(ndisp=24), dict(crad=3), dict(p1=200, p2=100), dict(thr=-1), dict(factor=0.0), dict(ndisp=512), dict(p2=5000), dict(wrad=6)):
        with pytest.raises(capi.BpvoError):
            ctx.stereo_bm(L[0], R[0], _sgm_params(ctx, **bad))
    # addFrame fed by the matcher on the device = addFrame fed the oracle's map from the host
    seq = synth.make_stereo_sequence(240, 320, 4, index=9)
    pp = hip.default_params(); pp.numPyramidLevels = 3; pp.verbosity = capi.VERB_SILENT; pp.descriptor = capi.DESC_BITPLANES
    a = hip.create(seq["K"], seq["b"], 240, 320, pp, n_frames=3, n_pairs=1)
    c = hip.create(seq["K"], seq["b"], 240, 320, pp, n_frames=3, n_pairs=1)
    sp = _sgm_params(a, ndisp=32)
    for left, right in seq["frames"]:
        ra = a.add_frame_stereo(left, right, sp)
        rc = c.add_frame(left, orc_sgm(orc, left, right, ndisp=32))
        assert np.array_equal(ra["pose"].view(np.uint32), rc["pose"].view(np.uint32)) and ra["isKeyFrame"] == rc["isKeyFrame"] and ra["stats"] == rc["stats"]


# ---- semi-global block matching: cv::StereoSGBM of OpenCV 2.4 as the reference constructs it (utils/stereo_algorithm.cc:27-40), BPVO_STEREO_SGBM ----
SGBM_DEFAULT = dict(mind=0, ndisp=64, wsz=3, p1=0, p2=0, d12=0, cap=0, uniq=0, spw=0, spr=0, fulldp=0)      # StereoSGBM(minD, nD, SADWindowSize, 0, 0, 0, 0, 0, 0, 0, false)


def orc_sgbm(orc, left, right, **kw):
    q = dict(SGBM_DEFAULT, **kw)
    out = np.empty(left.shape, np.float32)
    prm = (C.c_int * 11)(q["mind"], q["ndisp"], q["wsz"], q["p1"], q["p2"], q["d12"], q["cap"], q["uniq"], q["spw"], q["spr"], q["fulldp"])
    rc = orc.fn("stereo_sgbm")(np.ascontiguousarray(left).ctypes.data_as(C.c_void_p), np.ascontiguousarray(right).ctypes.data_as(C.c_void_p),
                               left.shape[0], left.shape[1], prm, out.ctypes.data_as(C.c_void_p))
    return out if rc == 0 else None


def np_sgbm(Limg, Rimg, mind=0, ndisp=64, wsz=3, p1=0, p2=0, d12=0, cap=0, uniq=0, spw=0, spr=0, fulldp=0):
    """cv::StereoSGBM (OpenCV 2.4, single-pass mode) + medianBlur(3) + filterSpeckles + / 16, evaluated from its definition with whole-array
    operations: Birchfield-Tomasi pixel costs on the clipped x-Sobel plane and (a quarter of) the raw plane, clamped SAD window with the two
    rows / columns the original never refreshes, five path recurrences (each one a plain scan over all its lines at once), sum, winner takes
    all, uniqueness, right-view voting, sub-pixel parabola, left-right check."""
    Li, Ri = Limg.astype(np.int64), Rimg.astype(np.int64)
    H, W = Li.shape
    minD, maxD, D = mind, mind + ndisp, ndisp
    SW = wsz if wsz > 0 else 5
    ft = max(cap, 15) | 1
    uniq = uniq if uniq >= 0 else 10
    d12 = d12 if d12 > 0 else 1
    P1 = p1 if p1 > 0 else 2
    P2 = max(p2 if p2 > 0 else 5, P1 + 1)
    minX1, maxX1 = max(maxD, 0), W + min(minD, 0)
    W1 = maxX1 - minX1
    INV = (minD - 1) * 16
    out = np.full((H, W), INV, np.int64)
    if W1 > 0:
        def planes(I):
            up, dn = I[np.maximum(np.arange(H) - 1, 0)], I[np.minimum(np.arange(H) + 1, H - 1)]
            sob = np.full((H, W), ft, np.int64)
            g = lambda A: A[:, 2:] - A[:, :-2]
            sob[:, 1:-1] = np.clip(2 * g(I) + g(up) + g(dn), -ft, ft) + ft
            raw = I.copy(); raw[:, 0] = ft; raw[:, -1] = ft
            return sob, raw

        def interval(P):      # min / max over the half-sample neighbourhood, the image edge replicated
            l = np.concatenate([P[:, :1], (P[:, 1:] + P[:, :-1]) // 2], axis=1)
            r = np.concatenate([(P[:, :-1] + P[:, 1:]) // 2, P[:, -1:]], axis=1)
            return np.minimum(np.minimum(l, r), P), np.maximum(np.maximum(l, r), P)

        pix = np.zeros((H, W1, D), np.int64)
        xs = np.arange(minX1, maxX1)
        for k, (pl, pr) in enumerate(zip(planes(Li), planes(Ri))):
            u0, u1 = interval(pl)
            v0, v1 = interval(pr)
            for d in range(D):
                xr = xs - (d + minD)
                u, v = pl[:, xs], pr[:, xr]
                c0 = np.maximum(np.maximum(0, u - v1[:, xr]), v0[:, xr] - u)
                c1 = np.maximum(np.maximum(0, v - u1[:, xs]), u0[:, xs] - v)
                pix[:, :, d] += np.minimum(c0, c1) >> (0 if k == 0 else 2)
        s2 = SW // 2
        xi = np.clip(np.arange(W1)[:, None] + np.arange(-s2, s2 + 1)[None, :], 0, W1 - 1)
        hs = pix[:, xi, :].sum(axis=2)
        yi = np.clip(np.arange(H)[:, None] + np.arange(-s2, s2 + 1)[None, :], 0, H - 1)
        Cv = hs[yi].sum(axis=1)
        Cv[max(H - s2, 1):] = Cv[max(H - s2, 1) - 1]             # the rows whose window would pass the last image row keep the last cost computed
        Cv[1:, 0, :] = Cv[0, 0, :]                               # the first cost column is only ever computed for row 0
        Cv = ((Cv + 32768) % 65536) - 32768                      # int16 buffers

        def recur(Lp, mp, Cc):
            big = np.full(Lp.shape[:-1] + (1,), 1 << 20, np.int64)
            lm = np.concatenate([big, Lp[..., :-1]], axis=-1) + P1
            lp = np.concatenate([Lp[..., 1:], big], axis=-1) + P1
            delta = (mp + P2)[..., None]
            Lv = Cc + np.minimum(np.minimum(Lp, lm), np.minimum(lp, delta)) - delta
            return ((Lv + 32768) % 65536) - 32768, Lv.min(axis=-1)

        def horizontal(step):
            Lo = np.zeros((H, W1, D), np.int64)
            Lp, mp = np.zeros((H, D), np.int64), np.zeros(H, np.int64)
            for x in (range(W1) if step > 0 else range(W1 - 1, -1, -1)):
                Lp, mp = recur(Lp, mp, Cv[:, x, :])
                Lo[:, x, :] = Lp
            return Lo

        def from_above(dx):      # predecessor (x + dx, y - 1); outside the buffer: L = 0, min L = 0
            Lo = np.zeros((H, W1, D), np.int64)
            Lp, mp = np.zeros((W1, D), np.int64), np.zeros(W1, np.int64)
            for y in range(H):
                if dx:
                    Ls, ms = np.zeros_like(Lp), np.zeros_like(mp)
                    if dx < 0: Ls[1:], ms[1:] = Lp[:-1], mp[:-1]
                    else: Ls[:-1], ms[:-1] = Lp[1:], mp[1:]
                else:
                    Ls, ms = Lp, mp
                Lp, mp = recur(Ls, ms, Cv[y])
                Lo[y] = Lp
            return Lo

        S = np.clip(horizontal(+1) + from_above(-1) + from_above(0) + from_above(+1), -32768, 32767)
        S = np.clip(S + horizontal(-1), -32768, 32767)
        best = S.argmin(axis=2)
        minS = S.min(axis=2)
        dd = np.arange(D)[None, None, :]
        unique = ~((S * (100 - uniq) < (minS * 100)[..., None]) & (np.abs(best[..., None] - dd) > 1)).any(axis=2)
        cdiv = lambda a, b: np.sign(a) * (np.abs(a) // b)        # C integer division (b > 0)
        bm, bp = np.clip(best - 1, 0, D - 1), np.clip(best + 1, 0, D - 1)
        Sm, Sp_, S0 = np.take_along_axis(S, bm[..., None], 2)[..., 0], np.take_along_axis(S, bp[..., None], 2)[..., 0], minS
        den = np.maximum(Sm + Sp_ - 2 * S0, 1)
        sub = np.where((best > 0) & (best < D - 1), best * 16 + cdiv((Sm - Sp_) * 16 + den, den * 2), best * 16) + minD * 16
        for y in range(H):
            d2, c2 = np.full(W, INV, np.int64), np.full(W, 32767, np.int64)
            for x in range(W1 - 1, -1, -1):
                if not unique[y, x]:
                    continue
                x2 = x + minX1 - best[y, x] - minD
                if c2[x2] > minS[y, x]:
                    c2[x2], d2[x2] = minS[y, x], best[y, x] + minD
                out[y, x + minX1] = sub[y, x]
            for x in range(minX1, maxX1):
                v = out[y, x]
                if v == INV:
                    continue
                lo, hi = v >> 4, (v + 15) >> 4
                xa, xb = x - lo, x - hi
                if 0 <= xa < W and d2[xa] >= minD and abs(d2[xa] - lo) > d12 and 0 <= xb < W and d2[xb] >= minD and abs(d2[xb] - hi) > d12:
                    out[y, x] = INV
    pad = np.pad(out, 1, mode="edge")
    out = np.median(np.stack([pad[1 + dy: 1 + dy + H, 1 + dx: 1 + dx + W] for dy in (-1, 0, 1) for dx in (-1, 0, 1)]), axis=0).astype(np.int64)
    if spw > 0:
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        idx = np.arange(H * W).reshape(H, W)
        ok = out != INV
        md = 16 * spr
        eh = ok[:, :-1] & ok[:, 1:] & (np.abs(out[:, :-1] - out[:, 1:]) <= md)
        ev = ok[:-1] & ok[1:] & (np.abs(out[:-1] - out[1:]) <= md)
        a = np.concatenate([idx[:, :-1][eh], idx[:-1][ev]]); b = np.concatenate([idx[:, 1:][eh], idx[1:][ev]])
        _, lab = connected_components(coo_matrix((np.ones(a.size), (a, b)), shape=(H * W, H * W)), directed=False)
        size = np.bincount(lab)
        out = np.where(ok & (size[lab].reshape(H, W) <= spw), INV, out)
    return (out.astype(np.float32) * np.float32(1.0 / 16.0)).astype(np.float32)


def _sgbm_pair(rows, cols, seed, z0=4.0):
    d = synth.make_stereo_pair(rows, cols, seed, z0=z0)
    rng = np.random.default_rng(seed)
    left, right = d["left"].copy(), d["right"].copy()
    right[: rows // 5] = rng.integers(0, 256, (rows // 5, cols), dtype=np.uint8)      # a band where nothing matches: uniqueness, left-right check
    left[rows // 2: rows // 2 + 6, 20: 20 + cols // 4] = 100                         # a textureless patch
    right[rows // 2: rows // 2 + 6, 20: 20 + cols // 4] = 100
    return left, right, d


@pytest.mark.parametrize("rows,cols,kw", [(40, 90, dict(ndisp=16, wsz=7)),                                        # conf/kitti_seq_0.cfg's shape: window 7, everything else 0
                                          (33, 75, dict(ndisp=32, wsz=3, p1=8, p2=32, uniq=10, d12=1)),
                                          (36, 96, dict(ndisp=16, wsz=5, p1=24, p2=96, cap=31, uniq=5, spw=20, spr=2)),
                                          (30, 80, dict(ndisp=16, wsz=9, mind=3, p1=50, p2=200, uniq=15, d12=2, spw=1)),
                                          (28, 70, dict(ndisp=16, wsz=0, uniq=-1, d12=-1, cap=63))])
def test_sgbm_oracle_against_a_numpy_evaluation_of_the_definition(orc, rows, cols, kw):
    left, right, _ = _sgbm_pair(rows, cols, 7)
    got = orc_sgbm(orc, left, right, **kw)
    want = np_sgbm(left, right, **kw)
    assert got is not None and np.array_equal(got, want), (np.argwhere(got != want)[:8], got[got != want][:8], want[got != want][:8])
    inv = (kw.get("mind", 0) - 1)
    assert (got == inv).any() and (got > inv).mean() > 0.2


def test_sgbm_recovers_the_disparity_of_a_plane_and_rejects_what_is_not_restated(orc):
    rows, cols = 60, 200
    d = synth.make_stereo_pair(rows, cols, 3, z0=4.0)
    got = orc_sgbm(orc, d["left"], d["right"], ndisp=32, wsz=7, p1=8 * 49, p2=32 * 49, uniq=10)
    ok = got >= 0
    ok[:, :32] = False
    assert ok.mean() > 0.5 and np.abs(got[ok] - d["disp"][ok]).mean() < 0.6
    for bad in (dict(ndisp=24), dict(mind=-2), dict(fulldp=1), dict(ndisp=0)):
        assert orc_sgbm(orc, d["left"], d["right"], **dict(dict(ndisp=32, wsz=7), **bad)) is None, bad


def _hip_sgbm_params(ctx, **kw):
    """the cv::StereoSGBM fields, set directly"""
    q = dict(SGBM_DEFAULT, **kw)
    sp = ctx.default_stereo_params(q["ndisp"])
    sp.algorithm = capi.STEREO_SGBM
    sp.minDisparity, sp.SADWindowSize, sp.P1, sp.P2, sp.disp12MaxDiff = q["mind"], q["wsz"], q["p1"], q["p2"], q["d12"]
    sp.preFilterCap, sp.uniquenessRatio, sp.speckleWindowSize, sp.speckleRange, sp.fullDP = q["cap"], q["uniq"], q["spw"], q["spr"], q["fulldp"]
    return sp


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,kw", [(376, 1241, dict(ndisp=128, wsz=7)),                                   # conf/kitti_seq_0.cfg at KITTI size
                                          (480, 640, dict(ndisp=64, wsz=7)),
                                          (480, 640, dict(ndisp=64, wsz=5, p1=8 * 25, p2=32 * 25, uniq=10, d12=1, spw=100, spr=2, cap=63)),   # OpenCV's documented set-up
                                          (121, 163, dict(ndisp=32, wsz=3, p1=8, p2=32, uniq=10, d12=1)),
                                          (97, 203, dict(ndisp=48, wsz=9, mind=3, p1=50, p2=200, uniq=15, d12=2, spw=1)),
                                          (64, 300, dict(ndisp=256, wsz=5, p1=10, p2=120, uniq=5)),
                                          (50, 70, dict(ndisp=16, wsz=0, uniq=-1, d12=-1, cap=63)),
                                          (40, 90, dict(ndisp=16, wsz=11, p1=3, p2=4)),
                                          (33, 60, dict(ndisp=64, wsz=3))])                                      # no cost column: every pixel invalid
def test_hip_sgbm_bit_exact(hip, orc, rows, cols, kw):
    left, right, d = _sgbm_pair(rows, cols, 4, z0=8.0 if cols > 700 else 4.0)
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(d["K"], d["b"], rows, cols, p, n_frames=3, n_pairs=1)
    got = ctx.stereo_bm(left, right, _hip_sgbm_params(ctx, **kw))
    want = orc_sgbm(orc, left, right, **kw)
    assert want is not None
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (np.argwhere(got != want)[:8], got[got != want][:8], want[got != want][:8], (got != want).sum())
    inv = kw.get("mind", 0) - 1
    if cols > kw["ndisp"] + kw.get("mind", 0) + 8:
        cost_columns = (cols - kw["ndisp"] - kw.get("mind", 0)) / cols
        assert (got == inv).any() and (got > inv).mean() > 0.2 * cost_columns
    else:
        assert (got == inv).all()


@pytest.mark.gpu
def test_hip_sgbm_reference_constructor_call_batch_errors_and_add_frame(hip, orc):
    """The reference's own construction (utils/stereo_algorithm.cc:30-39): nine positional arguments into cv::StereoSGBM's eleven-argument
    constructor — the config keys land one slot off.  bpvo_hip_stereo_params_sgbm_from_config reproduces it; batches; what the device
    path refuses; addFrame fed by the matcher."""
    rows, cols, n = 120, 200, 3
    pairs = [synth.make_stereo_pair(rows, cols, k, z0=4.0) for k in range(n)]
    L = np.stack([q["left"] for q in pairs]); R = np.stack([q["right"] for q in pairs])
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(pairs[0]["K"], pairs[0]["b"], rows, cols, p, n_frames=3, n_pairs=1)
    # keys of a config: uniquenessRatio = 12 -> disp12MaxDiff, speckleWindowSize = 40 -> preFilterCap, speckleRange = 7 -> uniquenessRatio, fullDP = 1 -> speckleWindowSize 1
    sp = ctx.sgbm_params_from_config(0, 32, SADWindowSize=5, P1=20, P2=90, uniquenessRatio=12, speckleWindowSize=40, speckleRange=7, fullDP=1)
    assert (sp.algorithm, sp.disp12MaxDiff, sp.preFilterCap, sp.uniquenessRatio, sp.speckleWindowSize, sp.speckleRange, sp.fullDP) == (capi.STEREO_SGBM, 12, 40, 7, 1, 0, 0)
    got = ctx.stereo_bm(L, R, sp)
    for k in range(n):
        assert np.array_equal(got[k], orc_sgbm(orc, L[k], R[k], ndisp=32, wsz=5, p1=20, p2=90, d12=12, cap=40, uniq=7, spw=1, spr=0)), k
    # conf/kitti_seq_0.cfg: minDisparity 0, numberOfDisparities 128 (here 32), SADWindowSize 7, fullDP 0, nothing else
    sp0 = ctx.sgbm_params_from_config(0, 32, SADWindowSize=7)
    assert np.array_equal(ctx.stereo_bm(L[0], R[0], sp0), orc_sgbm(orc, L[0], R[0], ndisp=32, wsz=7))
    for bad in (dict(ndisp=24), dict(mind=-1), dict(fulldp=1), dict(ndisp=512), dict(wsz=21), dict(wsz=7, p2=32000), dict(ndisp=0)):
        with pytest.raises(capi.BpvoError):
            ctx.stereo_bm(L[0], R[0], _hip_sgbm_params(ctx, **dict(dict(ndisp=32, wsz=7), **bad)))
    seq = synth.make_stereo_sequence(240, 320, 4, index=9)
    pp = hip.default_params(); pp.numPyramidLevels = 3; pp.verbosity = capi.VERB_SILENT; pp.descriptor = capi.DESC_BITPLANES
    a = hip.create(seq["K"], seq["b"], 240, 320, pp, n_frames=3, n_pairs=1)
    c = hip.create(seq["K"], seq["b"], 240, 320, pp, n_frames=3, n_pairs=1)
    sps = a.sgbm_params_from_config(0, 32, SADWindowSize=7)
    for left, right in seq["frames"]:
        ra = a.add_frame_stereo(left, right, sps)
        rc = c.add_frame(left, orc_sgbm(orc, left, right, ndisp=32, wsz=7))
        assert np.array_equal(ra["pose"].view(np.uint32), rc["pose"].view(np.uint32)) and ra["isKeyFrame"] == rc["isKeyFrame"] and ra["stats"] == rc["stats"]
