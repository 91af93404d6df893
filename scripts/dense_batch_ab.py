#!/usr/bin/env python3
"""Small batches of DENSE pairs (640x480, NMS off as conf/tsukuba.cfg: 300 k-point templates) on the team kernel against the four-kernel chain:
ms per step of bpvo_hip_batch_run with the option strings of argv[1:] (default: "", "team=0" and "persist_max_points=1000000": the team kernel whatever the templates).   python scripts/dense_batch_ab.py [options ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()      # (before the library loads: both initialise the HIP runtime)
from bpvo_amd import capi, synth
import bpvo_amd
hip = bpvo_amd.load()
opts = sys.argv[1:] or ["", "team=0", "persist_max_points=1000000"]
cases = ((capi.DESC_INTENSITY, "640x480 NMS off, intensity / CubicHermite", 3, 480, 640, 0, 3), (capi.DESC_BITPLANES, "640x480 NMS off, bit-planes / kLinear", 0, 480, 640, 0, 3),
         (capi.DESC_INTENSITY, "1241x376 default NMS, intensity / kLinear", 0, 376, 1241, 1, 4))
for desc, dn, interp, rows, cols, nms, levels in cases:
    for n in (4, 16, 64):
        b = synth.make_batch(rows, cols, n, first_index=0, workers=8)
        d_i, d_d = torch.from_numpy(b["images"]).cuda(), torch.from_numpy(b["disparities"]).cuda()
        ref = None
        for o in opts:
            os.environ["BPVO_HIP_OPTIONS"] = o
            p = hip.default_params(); p.numPyramidLevels = levels; p.descriptor = desc; p.lossFunction = capi.LOSS_HUBER; p.verbosity = capi.VERB_SILENT
            if nms == 0: p.nonMaxSuppRadius = 0; p.minSaliency = 0.001
            p.interp = interp; p.maxIterations = 55
            p.parameterTolerance = 1e-6; p.functionTolerance = 1e-4; p.gradientTolerance = 1e-6; p.relaxTolerancesForCoarseLevels = 0
            ctx = hip.create(b["K"], b["b"], rows, cols, p, device=0, n_frames=2 * n, n_pairs=n)
            poses, _ = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): poses, stats = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            if ref is None: ref = poses.copy()
            print(f"{dn}, {n} pairs, options [{o}]: {1e3 * dt:.2f} ms per step, {ctx.total_linearizations() / 4 / dt / 1e3:.1f} k GN it/s, same poses as the first: {np.array_equal(ref.view(np.uint32), poses.view(np.uint32))}", flush=True)
            ctx.close()
