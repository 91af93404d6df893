"""Throughput of the stereo front-end kernels alone (block matching and SGM) on batches resident in HBM — the `stereo front-end` lines of
bench.py's other_configs, for use under rocprofv3 (`rocprofv3 --kernel-trace --stats -- python3 scripts/stereo_bench.py`)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bpvo_amd import synth
import bench

def stack(rows, cols, n):
    ps = [synth.make_stereo_pair(rows, cols, i) for i in range(n)]
    return (np.stack([q["left"] for q in ps]), np.stack([q["right"] for q in ps]), ps[0]["K"], ps[0]["b"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
k, t = stack(376, 1241, n), stack(480, 640, n)
seq = synth.make_stereo_sequence(480, 640, 9, index=23)
import torch
import bpvo_amd
from bpvo_amd import capi
# BPVO_AB_LIB: an experimental build of the library (scripts/build_exp.sh) instead of the product's
hip = capi.Binding(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["BPVO_AB_LIB"]), "bpvo_hip_") if os.environ.get("BPVO_AB_LIB") else bpvo_amd.load()
dev = torch.device("cuda", 0)
torch.cuda.init()
out = bench.stereo_lines(hip, torch, dev, 0, {"batches": {"block matching 1241x376 / 128": (376, 1241, 128, k, "bm"), "block matching 640x480 / 64": (480, 640, 64, t, "bm"),
                                                           "SGM 1241x376 / 128": (376, 1241, 128, k, "sgm"), "SGM 640x480 / 64": (480, 640, 64, t, "sgm"),
                                                           "SGBM 1241x376 / 128": (376, 1241, 128, k, "sgbm"), "SGBM 640x480 / 64": (480, 640, 64, t, "sgbm")},
                                               "sequence": (480, 640, 64, seq["frames"], seq["K"], seq["b"])})
for name, v in out.items():
    print(name, json.dumps(v))
