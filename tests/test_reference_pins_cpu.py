"""Pins of the oracle against the reference's OWN code, for the few pieces of the path that compile without Eigen / OpenCV
(oracle/Makefile target `ref` -> oracle/_ref/libbpvo_ref.so, built from the sources under /root/reference):
median() (bpvo/utils.h:224-252), the v128 byte operators the census is made of (bpvo/v128.h, bpvo/census.cc:42-57) and the
ConfigFile reader (bpvo/config_file.cc).  Skipped where neither the prebuilt library nor /root/reference exists."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from util import bits_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libbpvo_ref.so")
REF_TREE = "/root/reference"


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(REF_LIB):
        if not os.path.isdir(os.path.join(REF_TREE, "bpvo")):
            pytest.skip("no oracle/_ref build and no reference tree")
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    lib = C.CDLL(REF_LIB)
    lib.ref_median.restype = C.c_float
    lib.ref_median.argtypes = [C.c_void_p, C.c_size_t]
    lib.ref_config_get.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    return lib


def test_median_rule_matches_the_reference(ref, orc):
    """bpvo::median: n < 3 -> data[0]; odd -> middle; even -> (max(lower half) + middle) / 2.0 in double (Q5)."""
    lib = orc.lib
    lib.bpvo_orc_median.restype = C.c_float
    lib.bpvo_orc_median.argtypes = [C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 17, 64, 255, 256, 1000, 4097, 50000]
    for n in sizes:
        for kind in range(4):
            if kind == 0:
                v = rng.standard_normal(n).astype(np.float32)
            elif kind == 1:
                v = np.abs(rng.standard_normal(n)).astype(np.float32) * 1e-3
            elif kind == 2:
                v = rng.integers(0, 4, n).astype(np.float32)            # many ties
            else:
                v = np.sort(rng.random(n).astype(np.float32))[::-1].copy()
            a = lib.bpvo_orc_median(v.ctypes.data_as(C.c_void_p), n)
            b = ref.ref_median(v.ctypes.data_as(C.c_void_p), n)
            assert np.float32(a).tobytes() == np.float32(b).tobytes(), (n, kind, a, b)
    assert ref.ref_median(None, 0) == 0.0 and lib.bpvo_orc_median(None, 0) == 0.0      # empty -> 0 (with a warning)


def test_census_comparisons_match_the_reference_operators(ref, orc):
    """bit k of the census byte = [neighbour_k >= centre] with the reference's v128 operator>= (unsigned bytes)."""
    rng = np.random.default_rng(12)
    a = rng.integers(0, 256, (200, 16), dtype=np.uint8)
    b = rng.integers(0, 256, (200, 16), dtype=np.uint8)
    b[:50] = a[:50]                                   # equality
    a[50:60] = 0; b[60:70] = 255                      # extremes (signed-compare traps: 0x80.. vs 0x7f..)
    out = np.empty(16, np.uint8)
    for i in range(len(a)):
        ref.ref_v128_ge(a[i].ctypes.data_as(C.c_void_p), b[i].ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out, np.where(a[i] >= b[i], 255, 0).astype(np.uint8))
    # a whole image through censusOp's composition of those operators against the oracle's census
    rows, cols = 37, 83
    img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    img[10:20, 10:40] = 128                           # flat region: all neighbours equal -> 0xff
    want = np.zeros((rows, cols), np.uint8)
    orc.lib.bpvo_orc_census(img.ctypes.data_as(C.c_void_p), rows, cols, C.c_float(-1.0), want.ctypes.data_as(C.c_void_p))
    got = np.zeros((rows, cols), np.uint8)
    offs = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
    pad = np.pad(img, ((1, 1), (1, 16)), mode="edge")
    for y in range(1, rows - 1):
        for x0 in range(1, cols - 1, 16):
            nbr = np.stack([pad[y + 1 + dy, x0 + 1 + dx: x0 + 1 + dx + 16] for dy, dx in offs]).copy()
            ctr = pad[y + 1, x0 + 1: x0 + 17].copy()
            o = np.empty(16, np.uint8)
            ref.ref_census_bytes(nbr.ctypes.data_as(C.c_void_p), ctr.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p))
            w = min(16, cols - 1 - x0)
            got[y, x0: x0 + w] = o[:w]
    assert np.array_equal(got, want)
    assert want[15, 20] == 0xff and not want[0].any() and not want[:, 0].any() and not want[:, -1].any()


def _facade_get(exe, path, key, default):
    out = subprocess.run([exe, path, "get", key, default], capture_output=True, text=True)
    return out.returncode, out.stdout.strip()


def test_config_file_reader_matches_the_reference(ref, tmp_path):
    """include/bpvo_hip/config_file.hpp against bpvo::ConfigFile on the reference's own conf/*.cfg files and on edge cases."""
    exe = str(tmp_path / "config_file_test")
    csrc = os.path.join(ROOT, "bpvo_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "config_file_test.cc"),
                        "-o", exe, "-L", csrc, "-lbpvo_hip", f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    files = sorted(glob.glob(os.path.join(REF_TREE, "conf", "*.cfg")))
    edge = tmp_path / "edge.cfg"
    edge.write_text("# comment\n% other comment\n\n  Descriptor   =   BitPlanes  \nMAXITERATIONS=7\nsigmaBitPlanes = 1.5 # trailing\nname = a b c\n")
    files.append(str(edge))
    keys = ["Descriptor", "descriptor", "DESCRIPTOR", "maxIterations", "lossFunction", "numPyramidLevels", "sigmaBitPlanes", "StereoAlgorithm",
            "Interpolation", "GradientEstimation", "Verbosity", "minSaliency", "name", "noSuchKey", "DataSet", "numberOfDisparities"]
    buf = C.create_string_buffer(1024)
    checked = 0
    for f in files:
        for k in keys:
            rc = ref.ref_config_get(f.encode(), k.encode(), b"DEFAULT", buf, 1024)
            frc, fval = _facade_get(exe, f, k, "DEFAULT")
            if rc == 2:
                assert frc != 0, (f, k, buf.value)
            else:
                assert frc == 0 and fval == buf.value.decode(), (f, k, fval, buf.value)
                checked += 1
    assert checked >= len(keys)
    for bad in ("novalue\n", "a = b = c\n", "= 3\n", "a =\n", "x = 1\n   \ny = 2\n", "a == 3\n", "\t\n", "#c\n a = 4"):
        p = tmp_path / "bad.cfg"
        p.write_text(bad)
        rc = ref.ref_config_get(str(p).encode(), b"a", b"D", buf, 1024)
        frc, fval = _facade_get(exe, str(p), "a", "D")
        assert (rc == 2) == (frc != 0), (bad, rc, buf.value, frc, fval)
    assert ref.ref_icompare(b"BitPlanes", b"bitplanes") == 1 and ref.ref_icompare(b"Bit", b"Bits") == 0


def test_simd_dot_and_abs_match_the_restated_forms(ref):
    """simd::dot (bpvo/simd.h:69-80, _mm_dp_ps under SSE4.1) = (a0 b0 + a1 b1) + (a2 b2 + a3 b3) in f32 — the summation order
    the projectPoints / BilinearInterp formulation is restated with; simd::abs clears the sign bit (-0.0 -> +0.0, NaN kept)."""
    ref.ref_simd_dot.restype = C.c_float
    rng = np.random.default_rng(13)
    for _ in range(2000):
        a = (rng.standard_normal(4) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        b = rng.random(4).astype(np.float32)
        got = np.float32(ref.ref_simd_dot(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        p = a * b
        want = np.float32(np.float32(p[0] + p[1]) + np.float32(p[2] + p[3]))
        assert got.tobytes() == want.tobytes(), (a, b, got, want)
    x = np.array([-0.0, -1.5, 2.25, -np.inf], np.float32)
    out = np.empty(4, np.float32)
    ref.ref_simd_abs(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert out.tobytes() == np.array([0.0, 1.5, 2.25, np.inf], np.float32).tobytes()


def test_weight_functions_agree_with_robust_loss_h(ref, orc):
    """bpvo/robust_loss.h:50-72 (Huber::weight, Tukey::weight: w(r / (sigma * k))) — the reference's header-only statement of
    the two M-estimator weights, the only one that compiles here — against the oracle's restatement of the SIMD bodies of
    bpvo/mestimator.cc:242-366 (w = k / max(|r / sigma|, k); (1 - (r / sigma / t)^2)^2 for |r / sigma| < t).  Different
    operation order (one reciprocal of sigma * k there, 1 / sigma then * (1 / t) here), so the bar is rounding: 4 ulp-ish."""
    ref.ref_huber_weight.restype = C.c_float
    ref.ref_huber_weight.argtypes = [C.c_float, C.c_float]
    ref.ref_tukey_weight.restype = C.c_float
    ref.ref_tukey_weight.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(21)
    n = 4096
    for sigma in (0.02, 0.37, 1.0, 3.5, 40.0):
        r = (rng.standard_normal(n) * sigma * rng.choice([0.05, 1.0, 3.0, 8.0], n)).astype(np.float32)
        r[:4] = [0.0, -0.0, sigma * 1.345, -sigma * 4.685]
        valid = np.ones(n, np.uint16)
        for loss, fn in ((0x10, ref.ref_huber_weight), (0x11, ref.ref_tukey_weight)):
            w = np.empty(n, np.float32)
            assert orc.lib.bpvo_orc_compute_weights(loss, r.ctypes.data_as(C.c_void_p), valid.ctypes.data_as(C.c_void_p), C.c_size_t(n),
                                                    C.c_float(sigma), w.ctypes.data_as(C.c_void_p)) == 0
            want = np.array([fn(sigma, float(x)) for x in r], np.float32)
            # at the Tukey cut-off the two forms may disagree on which side of t a residual falls: compare away from it
            edge = np.abs(np.abs(r / np.float32(sigma)) - (4.685 if loss == 0x11 else 1.345)) < 1e-4
            assert np.all(np.abs(w - want)[~edge] <= 1e-6), (sigma, loss, np.abs(w - want)[~edge].max())
            assert np.all((w >= 0) & (w <= 1))


def test_the_fuzz_rule_solver_acceptance_flip_on_synthetic_traces():
    """tests/tools/fuzz_parity.py acceptance_flip_in_traces: the rule fires only when (a) the iterates agree up to one linearisation, (b) the two
    steps there differ by a factor and (c) the ORACLE'S solver returns each side's step from that side's (H, G).  Synthetic traces around a
    well-conditioned system: identical runs -> no; a step that differs by a factor but is NOT what the solver gives -> no; the oracle's side
    solved from a damped system, the other from the plain one, each consistent with its own (H, G) -> yes; the same after the poses parted -> no."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import ctypes as C
    import fuzz_parity as fz
    import __graft_entry__ as ge
    from bpvo_amd import capi
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    rng = np.random.default_rng(7)
    A = rng.normal(size=(6, 6))
    H = (A @ A.T + 6.0 * np.eye(6)).astype(np.float32)
    G = rng.normal(size=6).astype(np.float32)

    def solve(Hm, Gv):
        dp = np.zeros(6, np.float32)
        Hc, Gc = np.ascontiguousarray(Hm.reshape(-1), np.float32), np.ascontiguousarray(Gv, np.float32)
        assert orc.fn("solve")(Hc.ctypes.data_as(C.c_void_p), Gc.ctypes.data_as(C.c_void_p), dp.ctypes.data_as(C.c_void_p))
        return dp

    def record(Hm, Gv, dp, level=0, T=None):
        r = np.zeros(68, np.float32)
        r[:16] = (np.eye(4, dtype=np.float32) if T is None else T).reshape(-1)
        r[16:52] = Hm.reshape(-1); r[52:58] = Gv; r[58] = 1.0; r[59] = 1.0; r[60] = 100.0; r[61:67] = dp; r[67] = level
        return r

    K = np.array([[500.0, 0, 80], [0, 500.0, 60], [0, 0, 1]], np.float32)
    dp = solve(H, G)
    same = np.stack([record(H, G, dp), record(H, G, dp)])
    assert not fz.acceptance_flip_in_traces(orc, same, same.copy(), K, 1.0)
    # a step ten times shorter that the solver does NOT give for that system
    short = np.stack([record(H, G, dp), record(H, G, 0.1 * dp)])
    assert not fz.acceptance_flip_in_traces(orc, short, same, K, 1.0)
    # a damped system on one side (what the f64 fallback amounts to: a much shorter step), each side consistent with its own (H, G)
    Hd = (H + 60.0 * np.eye(6, dtype=np.float32)).astype(np.float32)
    dpd = solve(Hd, G)
    assert np.linalg.norm(dp) > 5.0 * np.linalg.norm(dpd)
    gpu = np.stack([record(H, G, dp), record(H, G, dp)])
    cpu = np.stack([record(H, G, dp), record(Hd, G, dpd)])
    assert fz.acceptance_flip_in_traces(orc, gpu, cpu, K, 1.0)
    # ... but not when the iterates had already left the bar before that linearisation
    T_far = np.eye(4, dtype=np.float32); T_far[0, 3] = 0.05
    cpu_far = np.stack([record(H, G, dp, T=T_far), record(Hd, G, dpd, T=T_far)])
    assert not fz.acceptance_flip_in_traces(orc, gpu, cpu_far, K, 1.0)
