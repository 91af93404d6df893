"""The serial math of the gn_step kernel against the oracle, without a GPU: bpvo_amd/csrc/device_math.h is
__host__ __device__, so tests/cpp/host_math_harness.hip compiles the product's own source for the host and the results are
compared bit for bit with the oracle's restatement of the same reference lines (PoseEstimatorData_::solve with its
augmented f64 fallback, bpvo/pose_estimator_base.h:90-148; math::TwistToMatrix, bpvo/math_utils.h:140-168).  The
interesting inputs are the badly conditioned systems a pose estimation without Hartley normalisation produces: that is
where a pivoting or tolerance slip in the register-resident LDLT of the product would show."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    import __graft_entry__ as ge
    out = str(tmp_path_factory.mktemp("host_math") / "libhost_math.so")
    cmd = [ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "bpvo_amd", "csrc"), "-o", out, os.path.join(ROOT, "tests", "cpp", "host_math_harness.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return C.CDLL(out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_solver_matches_the_oracle_on_ill_conditioned_systems(harness, orc):
    rng = np.random.default_rng(0)
    fallbacks = 0
    for trial in range(20000):
        n = int(rng.integers(6, 40))
        scale = 10.0 ** rng.uniform(-4, 4, 6)                       # unnormalised columns: rotation vs translation units
        J = rng.standard_normal((n, 6)) * scale
        if rng.random() < 0.3:                                      # nearly dependent columns
            J[:, 5] = J[:, 4] * (1 + 1e-4 * rng.standard_normal()) + 1e-6 * rng.standard_normal(n) * scale[5]
        if rng.random() < 0.02:
            J[:, int(rng.integers(0, 6))] = 0.0                     # an unobservable parameter: zero pivot
        w = rng.random(n)
        H = np.ascontiguousarray((J.T * w) @ J, np.float32)
        H = np.ascontiguousarray(np.triu(H) + np.triu(H, 1).T, np.float32)     # symmetrised from the upper triangle, as toEigen does
        G = np.ascontiguousarray(J.T @ (w * rng.standard_normal(n)), np.float32)
        a, b = np.zeros(6, np.float32), np.zeros(6, np.float32)
        ra = orc.fn("solve")(_p(H), _p(G), _p(a))
        rb = harness.host_solve_system(_p(H), _p(G), _p(b))
        assert ra == rb and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (trial, ra, rb, a, b)
        # how often the augmented f64 path decided: the f32 solution must fail Eigen's isApprox test for that
        Hd, dpd = H.astype(np.float64), a.astype(np.float64)
        fallbacks += int(np.linalg.norm(Hd @ dpd - G) > 1e-5 * min(np.linalg.norm(Hd @ dpd), np.linalg.norm(G)) * 4)
    assert fallbacks > 100        # the fallback branch was exercised


def test_twist_exponential_matches_the_oracle(harness, orc):
    rng = np.random.default_rng(1)
    tw = orc.fn("twist_to_matrix", None)
    for trial in range(5000):
        mag = 10.0 ** rng.uniform(-10, 0.5)
        p = np.ascontiguousarray(rng.standard_normal(6) * mag, np.float32)
        if trial % 50 == 0:
            p[:3] = 0.0                                             # theta <= 1e-8: the pure-translation branch
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        tw(_p(p), _p(a))
        harness.host_twist_to_matrix(_p(p), _p(b))
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (trial, p)
