"""No-GPU checks of the drop-in boundary: the shared library loads, exports every symbol the header declares, fails
loudly without a device (no CPU fallback), and nothing in the product imports the oracle."""
import ctypes as C
import os
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
    for base in ("bpvo_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cc", ".cpp")):
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
    bad = tmp_path / "bad.cfg"
    bad.write_text("numPyramidLevels 3\n")
    out = subprocess.run([exe, str(bad)], capture_output=True, text=True)
    assert out.returncode == 1 and "Malformed ConfigFile line" in out.stdout
    out = subprocess.run([exe, str(tmp_path / "missing.cfg")], capture_output=True, text=True)
    assert out.returncode == 1 and "could not open file" in out.stdout
