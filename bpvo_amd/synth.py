"""Deterministic synthetic stereo pairs for the dense-alignment hot path.

A textured plane is rendered analytically (ray/plane intersection, no resampling of a raster) from two camera
poses.  Frame A sits at the identity pose and also gets its disparity map ``b*fx/Z``; frame B is rendered from
``T_gt`` (X_B = T_gt * X_A), which is the pose ``estimatePose(A, B)`` has to recover
(reference convention: bpvo/rigid_body_warp.h:111-121, x_cur = K * T * X_ref).

Calibrations: 640x480 -> fx=fy=615, c=(320,240), b=0.1 (reference: apps/vo_example.cc:60-61);
1241x376 -> KITTI seq-00 style fx=fy=718.856, c=(607.1928,185.2157), b=0.5372.
Everything is seeded: seed = 1000 + pair index.
"""
from __future__ import annotations

import numpy as np

CALIB = {
    (480, 640): dict(fx=615.0, fy=615.0, cx=320.0, cy=240.0, b=0.1),
    (376, 1241): dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, b=0.5372),
}


def calibration(rows: int, cols: int):
    """(K 3x3 float32, baseline) for an image size; unknown sizes scale the 640x480 calibration."""
    if (rows, cols) in CALIB:
        c = CALIB[(rows, cols)]
    else:
        s = cols / 640.0
        c = dict(fx=615.0 * s, fy=615.0 * s, cx=cols / 2.0, cy=rows / 2.0, b=0.1)
    K = np.array([[c["fx"], 0, c["cx"]], [0, c["fy"], c["cy"]], [0, 0, 1]], dtype=np.float32)
    return K, float(c["b"])


def _hash01(ix, iy, salt):
    """Integer lattice -> [0,1) (splitmix64-style mixing, vectorised)."""
    x = (ix.astype(np.int64) * np.int64(0x1F123BB5) + iy.astype(np.int64) * np.int64(0x5F356495) + np.int64(salt)).astype(np.uint64)
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _value_noise(u, v, cell, salt):
    """Bilinear value noise with lattice spacing `cell` (same units as u, v).

    The lattice hash is evaluated once on the bounding box of lattice nodes the image touches and gathered per pixel
    (identical values to hashing per pixel, ~4x faster for the 128-pair benchmark batches)."""
    fu, fv = u / cell, v / cell
    iu, iv = np.floor(fu), np.floor(fv)
    a, b = fu - iu, fv - iv
    iu, iv = iu.astype(np.int64), iv.astype(np.int64)
    u0, v0 = int(iu.min()), int(iv.min())
    gu, gv = np.meshgrid(np.arange(u0, int(iu.max()) + 2, dtype=np.int64), np.arange(v0, int(iv.max()) + 2, dtype=np.int64),
                         indexing="ij")
    table = _hash01(gu, gv, salt)
    ju, jv = iu - u0, iv - v0
    n00 = table[ju, jv]
    n10 = table[ju + 1, jv]
    n01 = table[ju, jv + 1]
    n11 = table[ju + 1, jv + 1]
    return (1 - b) * ((1 - a) * n00 + a * n10) + b * ((1 - a) * n01 + a * n11)


# lattice spacing in pixels at the plane's nominal depth, amplitude weight (sum 1)
OCTAVES = ((3.0, 0.30), (6.0, 0.25), (12.0, 0.20), (24.0, 0.15), (48.0, 0.10))


def _texture(u, v, seed, px_size):
    """Five octaves of value noise (lattice 3..48 px at the nominal depth), mapped to [16, 240].

    The fine octaves matter for the census-based descriptor: with only coarse (>= 8 px) bilinear cells the sign pattern of
    a 3x3 neighbourhood is constant inside a cell and the bit-planes carry almost no signal."""
    t = 0.0
    for k, (cell, wgt) in enumerate(OCTAVES):
        t = t + wgt * _value_noise(u, v, cell * px_size, seed * 7919 + 1 + k)
    return 16.0 + 224.0 * t


def twist_to_matrix(p):
    """SE(3) exponential, float64 (same formulas as bpvo/math_utils.h:140-168)."""
    p = np.asarray(p, dtype=np.float64)
    w, v = p[:3], p[3:]
    T = np.eye(4)
    th = np.linalg.norm(w)
    if th > 1e-8:
        a, b, ti = np.sin(th), 1 - np.cos(th), 1.0 / th
        S = ti * np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        S2 = S @ S
        T[:3, :3] = np.eye(3) + a * S + b * S2
        T[:3, 3] = (np.eye(3) + b * ti * S + (th - a) * ti * S2) @ v
    else:
        T[:3, 3] = v
    return T


def _render(K, b, rows, cols, T_cam_from_A, seed, z0, plane):
    """Render the plane Z = z0 + a*X + b*Y (frame-A coordinates) seen from a camera with X_cam = T * X_A."""
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    pa, pb = plane
    n = np.array([-pa, -pb, 1.0])          # n . X_A = z0
    Tinv = np.linalg.inv(T_cam_from_A)
    R, t = Tinv[:3, :3], Tinv[:3, 3]       # X_A = R * X_cam + t
    xs, ys = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    d_cam = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], axis=-1)
    d_A = d_cam @ R.T
    denom = d_A @ n
    s = (z0 - float(n @ t)) / denom        # X_cam = s * d_cam, so depth in the camera frame = s
    X_A = s[..., None] * d_A + t
    px_size = z0 / fx                      # metres per pixel at the nominal depth
    img = _texture(X_A[..., 0], X_A[..., 1], seed, px_size)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    disp = (b * fx / s).astype(np.float32)
    return img, disp


def make_pair(rows: int, cols: int, index: int = 0, max_rot: float = 0.01, max_trans: float = 0.05):
    """One synthetic pair. Returns dict(K, b, imgA, dispA, imgB, dispB, T_gt (float64 4x4), seed)."""
    seed = 1000 + int(index)
    rng = np.random.default_rng(seed)
    K, b = calibration(rows, cols)
    twist = np.concatenate([rng.uniform(-max_rot, max_rot, 3), rng.uniform(-max_trans, max_trans, 3)])
    T_gt = twist_to_matrix(twist)
    z0 = 10.0
    plane = (0.1, -0.15)
    imgA, dispA = _render(K, b, rows, cols, np.eye(4), seed, z0, plane)
    imgB, dispB = _render(K, b, rows, cols, T_gt, seed, z0, plane)
    return dict(K=K, b=b, imgA=imgA, dispA=dispA, imgB=imgB, dispB=dispB, T_gt=T_gt, seed=seed, twist=twist)


def make_stereo_pair(rows: int, cols: int, index: int = 0, z0: float = 10.0):
    """A rectified stereo pair of the plane scene: left image, right image (camera shifted by the baseline along +x), and the
    true disparity of the left image.  Returns dict(K, b, left, right, disp)."""
    seed = 1000 + int(index)
    K, b = calibration(rows, cols)
    plane = (0.1, -0.15)
    left, disp = _render(K, b, rows, cols, np.eye(4), seed, z0, plane)
    T_right = np.eye(4)
    T_right[0, 3] = -b                     # X_right = X_left - (b, 0, 0)
    right, _ = _render(K, b, rows, cols, T_right, seed, z0, plane)
    return dict(K=K, b=b, left=left, right=right, disp=disp, seed=seed)


def make_sequence(rows: int, cols: int, n_frames: int, index: int = 0, step_rot: float = 0.004, step_trans: float = 0.03):
    """A short camera trajectory over the same plane for addFrame tests: list of (img, disp) and absolute poses."""
    seed = 1000 + int(index)
    rng = np.random.default_rng(seed)
    K, b = calibration(rows, cols)
    T = np.eye(4)
    frames, poses = [], []
    for _ in range(n_frames):
        img, disp = _render(K, b, rows, cols, T, seed, 10.0, (0.1, -0.15))
        frames.append((img, disp))
        poses.append(T.copy())
        tw = np.concatenate([rng.uniform(-step_rot, step_rot, 3), rng.uniform(-step_trans, step_trans, 3)])
        T = twist_to_matrix(tw) @ T
    return dict(K=K, b=b, frames=frames, poses=poses)


def make_stereo_sequence(rows: int, cols: int, n_frames: int, index: int = 0, step_rot: float = 0.004, step_trans: float = 0.03):
    """make_sequence with the right image of every frame (the rig's right camera sits the baseline along +x of the left one):
    list of (left, right) and the true left disparities — input of the stereo front-end + addFrame."""
    seq = make_sequence(rows, cols, n_frames, index, step_rot, step_trans)
    seed = 1000 + int(index)
    shift = np.eye(4)
    shift[0, 3] = -seq["b"]
    frames = []
    for (img, _disp), T in zip(seq["frames"], seq["poses"]):
        right, _ = _render(seq["K"], seq["b"], rows, cols, shift @ T, seed, 10.0, (0.1, -0.15))
        frames.append((img, right))
    return dict(K=seq["K"], b=seq["b"], frames=frames, disps=[f[1] for f in seq["frames"]], poses=seq["poses"])


def _pair_for_batch(args):
    rows, cols, idx = args
    d = make_pair(rows, cols, idx)
    return d["imgA"], d["imgB"], d["dispA"], d["dispB"], d["T_gt"]


def make_batch(rows: int, cols: int, n_pairs: int, first_index: int = 0, workers: int = 1):
    """n_pairs pairs packed as the batch API wants them: images [2n, R, W] = A0,B0,A1,B1,..., disparities likewise.

    workers > 1 renders the pairs in a process pool (fork; call it before anything initialises the GPU)."""
    imgs = np.empty((2 * n_pairs, rows, cols), dtype=np.uint8)
    disps = np.empty((2 * n_pairs, rows, cols), dtype=np.float32)
    T_gt = np.empty((n_pairs, 4, 4), dtype=np.float64)
    K, b = calibration(rows, cols)
    jobs = [(rows, cols, first_index + p) for p in range(n_pairs)]
    if workers > 1 and n_pairs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, n_pairs)) as pool:
            results = pool.map(_pair_for_batch, jobs, chunksize=1)
    else:
        results = map(_pair_for_batch, jobs)
    for p, (ia, ib, da, db, tg) in enumerate(results):
        imgs[2 * p], imgs[2 * p + 1] = ia, ib
        disps[2 * p], disps[2 * p + 1] = da, db
        T_gt[p] = tg
    return dict(K=K, b=b, images=imgs, disparities=disps, T_gt=T_gt)
