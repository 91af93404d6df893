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


def test_batch_multi_gpu_cpp_matches_python(hip, tmp_path):
    """examples/batch_multi_gpu.cc on include/bpvo_hip/multi_gpu.h (one process, one context + host thread per GPU, one
    ncclGather of the records): on the GPUs present its gathered poses are bit for bit the ones bpvo_hip_batch_run gives
    through the ctypes binding.  The 1-GPU box exercises the whole path with a communicator of one rank; the driver's
    8-GPU node exercises the sharding."""
    import torch
    rows, cols, n, levels = 120, 160, 5, 3
    b = synth.make_batch(rows, cols, n, first_index=40)
    b["images"].tofile(tmp_path / "images.u8")
    b["disparities"].tofile(tmp_path / "disparities.f32")
    exe = str(tmp_path / "batch_multi_gpu")
    csrc = os.path.join(ROOT, "bpvo_amd", "csrc")
    assert os.path.exists(os.path.join(csrc, "libbpvo_hip_mgpu.so")), "run __graft_entry__.build() first"
    r = subprocess.run(["g++", "-std=c++11", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "batch_multi_gpu.cc"),
                        "-o", exe, "-L", csrc, "-lbpvo_hip_mgpu", "-lbpvo_hip", f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    K = b["K"]
    p = hip.default_params()
    p.numPyramidLevels = levels; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(K, b["b"], rows, cols, p, n_frames=2 * n, n_pairs=n)
    poses_py, stats_py = ctx.batch_run(b["images"], b["disparities"])
    for n_gpus in sorted({1, torch.cuda.device_count()}):
        prefix = str(tmp_path / f"out{n_gpus}")
        r = subprocess.run([exe, str(tmp_path), str(rows), str(cols), repr(float(K[0, 0])), repr(float(K[1, 1])), repr(float(K[0, 2])),
                            repr(float(K[1, 2])), repr(float(b["b"])), str(n), str(n_gpus), "bitplanes", str(levels), prefix],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert f"on {n_gpus} GPU(s)" in r.stdout
        poses_cpp = np.fromfile(prefix + "_poses.f32", np.float32).reshape(n, 4, 4)
        records = np.fromfile(prefix + "_records.f32", np.float32).reshape(n, 32)
        assert np.array_equal(poses_cpp.view(np.uint32), poses_py.view(np.uint32))
        from bpvo_amd.distributed import records_to_poses
        rp, it, _ = records_to_poses(records)
        assert np.array_equal(rp, poses_py) and np.array_equal(it[:, :levels], stats_py["numIterations"])


RAGGED_GATHER_SCRIPT = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as ge
import bpvo_amd
from bpvo_amd import capi, synth
hip = bpvo_amd.load()
lib = C.CDLL(ge.MGPU_LIB)
lib.bpvo_hip_node_last_error.restype = C.c_char_p
rt = C.CDLL("libamdhip64.so")
cnt = C.c_int()
assert rt.hipGetDeviceCount(C.byref(cnt)) == 0 and cnt.value >= 1
world = cnt.value
rows, cols, n, levels = 120, 160, 4, 2
b = synth.make_batch(rows, cols, n * world, first_index=7)
p = hip.default_params()
p.numPyramidLevels = levels; p.descriptor = capi.DESC_INTENSITY; p.lossFunction = capi.LOSS_HUBER; p.verbosity = capi.VERB_SILENT
node = C.c_void_p()
K = np.ascontiguousarray(b["K"], np.float32)
rc = lib.bpvo_hip_node_create(C.byref(node), world, None, K.ctypes.data_as(C.c_void_p), C.c_float(float(b["b"])), rows, cols, C.byref(p), n)
assert rc == 0, lib.bpvo_hip_node_last_error(None)
assert lib.bpvo_hip_node_num_devices(node) == world
poses = np.zeros((n * world, 4, 4), np.float32)
records = np.zeros((n * world, 32), np.float32)
rc = lib.bpvo_hip_node_batch_run(node, n * world, b["images"].ctypes.data_as(C.c_void_p), b["disparities"].ctypes.data_as(C.c_void_p),
                                 poses.ctypes.data_as(C.c_void_p), records.ctypes.data_as(C.c_void_p), None)
assert rc == 0, lib.bpvo_hip_node_last_error(node)
assert np.array_equal(records[:, :12].reshape(-1, 3, 4), poses[:, :3, :]) and np.all(poses[:, 3] == [0, 0, 0, 1])
# the same pairs through one plain context
ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2 * n * world, n_pairs=n * world)
ref, _ = ctx.batch_run(b["images"], b["disparities"])
assert np.array_equal(ref.view(np.uint32), poses.view(np.uint32))
# ragged: rank r contributes n - (r % 2) - 1 records
n_local = (C.c_int * world)(*[n - (r % 2) - 1 for r in range(world)])
out = np.zeros((sum(n_local), 32), np.float32)
rc = lib.bpvo_hip_gather_records(node, n_local, 0, out.ctypes.data_as(C.c_void_p))
assert rc == 0, lib.bpvo_hip_node_last_error(node)
expect = np.concatenate([records[r * n: r * n + n_local[r]] for r in range(world)])
assert np.array_equal(out.view(np.uint32), expect.view(np.uint32))
bad = (C.c_int * world)(*[n + 1] * world)
assert lib.bpvo_hip_gather_records(node, bad, 0, out.ctypes.data_as(C.c_void_p)) != 0
lib.bpvo_hip_node_destroy(node)
print("ragged gather ok on", world, "device(s)")
"""


def test_gather_records_with_ragged_blocks():
    """bpvo_hip_gather_records pads the blocks to the largest one for the single ncclGather and hands back exactly
    sum(n_local) records in rank order; the sharded batch equals the same pairs on one plain context bit for bit.  Runs in a
    fresh interpreter WITHOUT torch: libbpvo_hip_mgpu.so links /opt/rocm's librccl.so.1, and a process that has imported
    torch already holds torch's own bundled library under that name (multi_gpu.h says who should load which)."""
    import sys
    r = subprocess.run([sys.executable, "-c", RAGGED_GATHER_SCRIPT, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ragged gather ok" in r.stdout, r.stdout + r.stderr
