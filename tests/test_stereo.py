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
