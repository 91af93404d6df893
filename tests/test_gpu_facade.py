"""The C++ facade (include/bpvo_hip/vo.hpp) and the vo_perf harness (examples/vo_perf.cc, after apps/vo_perf.cc) on the GPU:
the C++ program must reproduce what the ctypes binding gets through the same C ABI."""
import os
import subprocess

import numpy as np
import pytest

from bpvo_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vo_perf_cpp_matches_python(hip, tmp_path):
    rows, cols, n = 120, 160, 6
    seq = synth.make_sequence(rows, cols, n, index=9, step_rot=0.004, step_trans=0.02)
    for i, (img, disp) in enumerate(seq["frames"]):
        img.tofile(tmp_path / f"image_{i:05d}.u8")
        disp.tofile(tmp_path / f"disparity_{i:05d}.f32")
    exe = str(tmp_path / "vo_perf")
    csrc = os.path.join(ROOT, "bpvo_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++11", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "vo_perf.cc"),
                        "-o", exe, "-L", csrc, "-lbpvo_hip", f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    K = seq["K"]
    prefix = str(tmp_path / "out")
    r = subprocess.run([exe, str(tmp_path), str(rows), str(cols), str(K[0, 0]), str(K[1, 1]), str(K[0, 2]), str(K[1, 2]), str(seq["b"]),
                        str(n), "intensity", prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    poses_cpp = np.loadtxt(prefix + "_poses.txt").reshape(-1, 4, 4)
    iters_cpp = np.loadtxt(prefix + "_iterations.txt").astype(int)
    assert poses_cpp.shape[0] == n

    # the same parameters as examples/vo_perf.cc through the ctypes binding
    p = hip.default_params()
    p.numPyramidLevels = 3; p.maxIterations = 100; p.parameterTolerance = 1e-6; p.functionTolerance = 1e-6
    p.verbosity = capi.VERB_SILENT; p.lossFunction = capi.LOSS_HUBER; p.descriptor = capi.DESC_INTENSITY
    p.minTranslationMagToKeyFrame = 0.1; p.minRotationMagToKeyFrame = 2.5
    p.maxFractionOfGoodPointsToKeyFrame = 0.7; p.goodPointThreshold = 0.8
    ctx = hip.create(K, seq["b"], rows, cols, p, n_frames=3, n_pairs=1)
    res = [ctx.add_frame(i, d) for i, d in seq["frames"]]
    for k in range(n):
        assert np.allclose(poses_cpp[k], res[k]["pose"], atol=2e-6), k          # text round trip of %g
        assert iters_cpp[k] == res[k]["stats"][0]["numIterations"]
    traj = np.loadtxt(prefix + "_path.txt")
    assert traj.shape == (n, 3) and np.allclose(traj, ctx.trajectory()[:, :3, 3], atol=2e-6)
