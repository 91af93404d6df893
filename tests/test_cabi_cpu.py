"""No-GPU checks of the drop-in boundary: the shared library loads, exports every symbol the header declares, fails
loudly without a device (no CPU fallback), and nothing in the product imports the oracle."""
import ctypes as C
import os

import numpy as np
import re
import subprocess

import pytest

import bpvo_amd
from bpvo_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bpvo_hip", "c_api.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bpvo_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_exports_every_declared_symbol():
    lib = bpvo_amd.load()      # raises if libbpvo_hip.so is missing
    names = declared_functions()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(lib.lib, n)]
    assert not missing, f"declared in c_api.h but not exported: {missing}"


def test_multi_gpu_library_exports_its_header_and_shards_like_the_python_helper():
    """include/bpvo_hip/multi_gpu.h (libbpvo_hip_mgpu.so: one ctx + host thread per GPU, one RCCL gather): every declared
    symbol is exported; the pure sharding rule — the only part that runs without a GPU — equals bpvo_amd.distributed's."""
    import __graft_entry__ as ge
    from bpvo_amd.distributed import shard_range
    path = ge.build_mgpu()
    lib = C.CDLL(path)
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bpvo_hip", "multi_gpu.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(bpvo_hip_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 8 and "bpvo_hip_gather_records" in names and "bpvo_hip_node_batch_run" in names
    assert not [n for n in names if not hasattr(lib, n)]
    lo, hi = C.c_int(), C.c_int()
    for n_total, world in ((1024, 8), (1024, 3), (5, 8), (0, 2), (7, 1)):
        covered = []
        for rank in range(world):
            lib.bpvo_hip_shard_range(n_total, rank, world, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == shard_range(n_total, rank, world)
            covered += list(range(lo.value, hi.value))
        assert covered == list(range(n_total))
    # no device here: creating a node fails loudly
    if not _has_gpu():
        lib.bpvo_hip_node_last_error.restype = C.c_char_p
        node = C.c_void_p()
        K = (C.c_float * 9)(500, 0, 80, 0, 500, 60, 0, 0, 1)
        p = bpvo_amd.load().default_params()
        rc = lib.bpvo_hip_node_create(C.byref(node), 1, None, K, C.c_float(0.1), 120, 160, C.byref(p), 2)
        assert rc != 0 and not node.value and lib.bpvo_hip_node_last_error(None)


def test_struct_layouts_match_header():
    src = open(HEADER).read()
    body = re.search(r"typedef struct bpvo_hip_params \{(.*?)\} bpvo_hip_params;", src, re.S).group(1)
    fields = re.findall(r"\b(?:int|float)\s+(\w+);", body)
    assert fields == [f[0] for f in capi.Params._fields_] and len(fields) == 35     # bpvo/types.h:171-413
    assert C.sizeof(capi.Params) == 35 * 4
    assert C.sizeof(capi.Stats) == 16
    assert C.sizeof(capi.Result) == 16 * 4 + 36 * 4 + 8 * 16 + 4 * 4
    assert capi.POINT_WITH_INFO.itemsize == 32                                      # bpvo/point_cloud.h:30-62


def test_default_params_equal_reference_defaults():
    lib = bpvo_amd.load()
    p = lib.default_params()
    # AlgorithmParameters() (reference: bpvo/types.cc:31-66)
    assert (p.numPyramidLevels, p.minImageDimensionForPyramid, p.maxIterations) == (-1, 40, 50)
    assert p.sigmaPriorToCensusTransform == -1.0 and p.sigmaBitPlanes == 0.5
    assert abs(p.parameterTolerance - 1e-7) < 1e-12 and abs(p.functionTolerance - 1e-6) < 1e-12 and abs(p.gradientTolerance - 1e-8) < 1e-13
    assert (p.gradientEstimation, p.interp, p.lossFunction, p.descriptor) == (capi.GRAD_CD3, capi.INTERP_LINEAR, capi.LOSS_TUKEY, capi.DESC_INTENSITY)
    assert (p.minNumPixelsForNonMaximaSuppression, p.nonMaxSuppRadius, p.maxTestLevel, p.withNormalization) == (76800, 1, 0, 1)
    assert abs(p.minSaliency - 0.1) < 1e-7 and abs(p.minValidDisparity - 0.001) < 1e-9 and p.maxValidDisparity == 512.0


def test_stereo_params_layout_and_the_references_sgbm_constructor_call():
    """bpvo_hip_stereo_params: the header's fields are the ctypes mirror's, in order; and bpvo_hip_stereo_params_sgbm_from_config fills the
    cv::StereoSGBM fields from the reference's config KEYS as its constructor call does — nine positional arguments into a constructor of
    eleven, so the keys land one slot off (utils/stereo_algorithm.cc:30-39): uniquenessRatio -> disp12MaxDiff, speckleWindowSize ->
    preFilterCap, speckleRange -> uniquenessRatio, (bool) fullDP -> speckleWindowSize; speckleRange and fullDP keep their defaults."""
    src = open(HEADER).read()
    body = re.search(r"typedef struct bpvo_hip_stereo_params \{(.*?)\} bpvo_hip_stereo_params;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int|double)\s+(\w+);", body)
    assert fields == [f[0] for f in capi.StereoParams._fields_]
    assert C.sizeof(capi.StereoParams) == 14 * 4 + 2 * 8 + 6 * 4
    lib = bpvo_amd.load()
    sp = capi.StereoParams()
    lib.fn("stereo_params_sgbm_from_config", None)(C.byref(sp), 2, 96, 9, 11, 22, 33, 44, 55, 1)
    assert (sp.algorithm, sp.minDisparity, sp.numberOfDisparities, sp.SADWindowSize, sp.P1, sp.P2) == (capi.STEREO_SGBM, 2, 96, 9, 11, 22)
    assert (sp.disp12MaxDiff, sp.preFilterCap, sp.uniquenessRatio, sp.speckleWindowSize, sp.speckleRange, sp.fullDP) == (33, 44, 55, 1, 0, 0)
    # conf/kitti_seq_0.cfg: minDisparity 0, numberOfDisparities 128, SADWindowSize 7, fullDP 0; the other keys at the defaults of their cf.get
    lib.fn("stereo_params_sgbm_from_config", None)(C.byref(sp), 0, 128, 7, 0, 0, 0, 0, 0, 0)
    assert (sp.P1, sp.P2, sp.disp12MaxDiff, sp.preFilterCap, sp.uniquenessRatio, sp.speckleWindowSize, sp.speckleRange, sp.fullDP) == (0,) * 8


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a box without a GPU")
def test_no_device_is_a_loud_error_not_a_fallback():
    import numpy as np
    lib = bpvo_amd.load()
    p = lib.default_params()
    p.numPyramidLevels = 2
    with pytest.raises(capi.BpvoError) as e:
        lib.create(np.eye(3, dtype=np.float32), 0.1, 64, 64, p)
    assert "-6" in str(e.value) and "no CPU fallback" in str(e.value)


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(bpvo_amd, "LIB_PATH", str(tmp_path / "libbpvo_hip.so"))
    with pytest.raises(RuntimeError):
        bpvo_amd.load()


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke()/build() and bench.py's cpu_baseline leg may reference oracle/."""
    offenders = []
    for base in ("bpvo_amd", "include", "examples", "scripts"):      # the parity tools that use the oracle live in tests/tools/
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cc", ".cpp", ".sh")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"bpvo_orc_|libbpvo_oracle|oracle/", txt):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
    out = subprocess.run(["ldd", bpvo_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_facade_header_compiles():
    """include/bpvo_hip/vo.hpp (the C++ facade with the reference's class names) is valid C++11 against the C ABI."""
    src = os.path.join(ROOT, "tests", "cpp", "facade_compile.cc")
    out = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def _build_cfg_tool(tmp_path):
    exe = str(tmp_path / "config_file_test")
    csrc = os.path.join(ROOT, "bpvo_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "config_file_test.cc"),
                        "-o", exe, "-L", csrc, "-lbpvo_hip", f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_config_file_reader_matches_reference_semantics(tmp_path):
    """conf/*.cfg style files through the facade (reference: bpvo/config_file.cc:50-71, bpvo/types.cc:68-107); the sample
    is the content of the reference's conf/perf_bitplanes.cfg parameters, retyped."""
    exe = _build_cfg_tool(tmp_path)
    cfg = tmp_path / "perf_bitplanes.cfg"
    cfg.write_text("# comment\n% another comment\nnumPyramidLevels = 3\nDescriptor = BitPlanes\n\nparameterTolerance = 1e-6\n"
                   "functionTolerance = 1e-4\n\nverbosity = Silent\n\nlossFunction = L2\n\nminTranslationMagToKeyFrame = 0.1\n"
                   "minRotationMagToKeyFrame = 5.0\n\nsigmaPriorToCensusTransform = 0.75\nsigmaBitPlanes = 1.6\n\nmaxIterations = 50\n"
                   "relaxTolerancesForCoarseLevels = 1\n")
    out = subprocess.run([exe, str(cfg)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    kv = dict(line.split() for line in out.stdout.strip().splitlines())
    assert kv["numPyramidLevels"] == "3" and kv["descriptor"] == str(capi.DESC_BITPLANES)      # case-insensitive key "Descriptor"
    assert kv["lossFunction"] == str(capi.LOSS_L2) and kv["verbosity"] == str(capi.VERB_SILENT)
    assert float(kv["parameterTolerance"]) == pytest.approx(1e-6) and float(kv["functionTolerance"]) == pytest.approx(1e-4)
    assert float(kv["sigmaPriorToCensusTransform"]) == 0.75 and float(kv["sigmaBitPlanes"]) == pytest.approx(1.6)
    # file defaults, not constructor defaults (bpvo/types.cc:68-107): CD5, gradientTolerance 1e-6, minValidDisparity 1
    assert kv["gradientEstimation"] == str(capi.GRAD_CD5) and float(kv["gradientTolerance"]) == pytest.approx(1e-6)
    assert float(kv["minValidDisparity"]) == 1.0 and float(kv["goodPointThreshold"]) == 0.75
    # printing helpers: labels, order and spelling of bpvo/types.cc:109-364
    out2 = subprocess.run([exe, str(cfg), "print"], capture_output=True, text=True)
    assert out2.returncode == 0, out2.stdout
    txt = out2.stdout.split("PRINT\n", 1)[1]
    blocks = txt.strip().split("\n--\n")
    assert blocks[0].splitlines()[0] == "numPyramidLevels = 3" and blocks[0].splitlines()[-1] == "maxTestLevel = 0"
    assert "gradienEstimation: CentralDifference_5" in blocks[0] and "InterpolationType: Linear" in blocks[0] and "lossFunction = L2" in blocks[0]
    assert "relaxTolerancesForCoarseLevel = 1" in blocks[0] and len(blocks[0].splitlines()) == 33
    assert blocks[1] == "numIterations: 0\nfinalError: -1\nfirstOrderOptimality: -1\nstatus: SolverError"
    assert blocks[2] == "[3,4]"
    assert blocks[3].splitlines()[0] == "1 0 0 0" and "isKeyFrame: false" in blocks[3] and blocks[3].endswith("status: SolverError")
    assert blocks[4] == "SmallFracOfGoodPoints CenteralDifference BitPlanes FunctionTolReached CubicHermite"
    bad = tmp_path / "bad.cfg"
    bad.write_text("numPyramidLevels 3\n")
    out = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert out.returncode == 1 and "Malformed ConfigFile line" in out.stdout
    out = subprocess.run([exe, str(tmp_path / "missing.cfg")], capture_output=True, text=True)
    assert out.returncode == 1 and "could not open file" in out.stdout


def test_io_formats_and_kitti_metric(tmp_path):
    """include/bpvo_hip/io.hpp: PLY (bpvo/point_cloud.cc:135-177), trajectory text (bpvo/trajectory.cc:53-97) and the KITTI
    segment-error metric (utils/kitti_eval.cc:80-167) against an independent numpy evaluation of the same files."""
    exe = str(tmp_path / "io_test")
    csrc = os.path.join(ROOT, "bpvo_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "io_test.cc"),
                        "-o", exe, "-L", csrc, "-lbpvo_hip", f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-500:])

    # PLY: header + 5 records of 3 x f32 + 4 x u8
    raw = (tmp_path / "cloud.ply").read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0" and lines[2] == "comment generated by bpvo"
    assert lines[3] == "comment io_test" and lines[4] == "element vertex 5"
    assert lines[5:] == ["property float x", "property float y", "property float z", "property uchar red", "property uchar green",
                         "property uchar blue", "property uchar alpha"]
    rec = np.frombuffer(body, dtype=np.dtype([("xyz", "<f4", 3), ("rgba", "u1", 4)]))
    assert len(rec) == 5 and np.allclose(rec["xyz"][3], [1.5, -3.0, 5.0]) and list(rec["rgba"][2]) == [20, 20, 20, 255]

    traj = np.loadtxt(tmp_path / "traj.txt").reshape(-1, 4, 4)
    path = np.loadtxt(tmp_path / "path.txt")
    gt = np.loadtxt(tmp_path / "gt" / "00.txt").reshape(-1, 3, 4)
    assert traj.shape[0] == 1200 and np.allclose(traj[:, :3, 3], path, atol=1e-3) and np.allclose(traj[:, :3, :], gt, atol=2e-3, rtol=1e-5)
    assert np.allclose(traj[:, 3], [0, 0, 0, 1])
    # the first pose is InvertPose(T) as written in the reference: R^T and -R t
    yaw = -0.002
    R = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
    assert np.allclose(traj[0, :3, :3], R.T, atol=1e-6) and np.allclose(traj[0, :3, 3], -R @ np.array([0, 0, -1.0]), atol=1e-6)

    # KITTI metric, second implementation
    def load(p):
        a = np.loadtxt(p).reshape(-1, 3, 4)
        T = np.tile(np.eye(4), (len(a), 1, 1))
        T[:, :3, :] = a
        return T

    def inv(T):      # utils/kitti_eval.cc:41-51 as written
        r = np.eye(4)
        r[:3, :3] = T[:3, :3].T
        r[:3, 3] = -T[:3, :3] @ T[:3, 3]
        return r

    G, E = load(tmp_path / "gt" / "00.txt"), load(tmp_path / "est" / "00.txt")
    d = np.concatenate([[0], np.cumsum(np.linalg.norm(np.diff(G[:, :3, 3], axis=0), axis=1))])
    want = []
    for f in range(0, len(G), 10):
        for L in (100, 200, 300, 400, 500, 600, 700, 800):
            idx = np.nonzero(d[f:] > d[f] + L)[0]
            if len(idx) == 0:
                continue
            fl = f + idx[0]
            Terr = inv(inv(E[f]) @ E[fl]) @ (inv(G[f]) @ G[fl])
            rr = np.arccos(np.clip(0.5 * (np.trace(Terr[:3, :3]) - 1), -1, 1))
            want.append((f, rr / L, np.linalg.norm(Terr[:3, 3]) / L, L, L / (0.1 * (fl - f + 1))))
    got = [tuple(float(x) for x in l.split()[1:]) for l in out.stdout.splitlines() if l.startswith("ERR")]
    assert len(got) == len(want) > 100
    got, want = np.array(got), np.array(want)
    assert np.array_equal(got[:, 0], want[:, 0]) and np.array_equal(got[:, 3], want[:, 3])
    assert np.allclose(got[:, 2], want[:, 2], rtol=2e-3, atol=1e-6) and np.allclose(got[:, 4], want[:, 4], rtol=1e-5)
    assert np.allclose(got[:, 1], want[:, 1], atol=5e-6)          # acos of a float near 1
    tl = np.loadtxt(tmp_path / "plot_tl.txt")
    # (the values are large because InvertPose, as written in the reference, is not the rigid inverse for R != I; the
    # metric is restated, not repaired)
    assert tl.shape[1] == 2 and tl[0, 0] == 100 and abs(tl[0, 1] - want[want[:, 3] == 100][:, 2].mean()) <= 2e-3 * tl[0, 1]
    assert (tmp_path / "plot_rs.txt").exists() and (tmp_path / "plot_ts.txt").stat().st_size > 0


def test_every_option_of_the_library_is_in_the_headers_table():
    """bpvo_hip_set_option's keys (context.hip option_table) against the table of include/bpvo_hip/c_api.h: a key without its line of
    default and meaning there is an undocumented switch."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bpvo_amd", "csrc", "context.hip")).read()
    table = src[src.index("const std::vector<OptionDef>& option_table()"):src.index("int set_option(")]
    keys = set(re.findall(r'OPT_INT\("([a-z_0-9]+)"', table)) | set(re.findall(r'OptionDef\{"([a-z_0-9]+)"', table))
    assert len(keys) >= 25, keys
    header = open(os.path.join(root, "include", "bpvo_hip", "c_api.h")).read()
    missing = sorted(k for k in keys if ' *   "%s"' % k not in header)
    assert not missing, missing
