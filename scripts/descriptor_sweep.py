#!/usr/bin/env python3
"""Cost per (template point x channel x linearisation) of a 64-pair 640x480 batch for every descriptor and interpolation type — a table in which a path
that falls off a cliff (a generic-channel kernel, a channel group, a missing tap cache) shows as an outlier.   python scripts/descriptor_sweep.py [pairs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from bpvo_amd import capi, synth
import bpvo_amd
hip = bpvo_amd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, cols, levels = 480, 640, 3
batch = synth.make_batch(rows, cols, n, first_index=0, workers=8)
d_i, d_d = torch.from_numpy(batch["images"]).cuda(), torch.from_numpy(batch["disparities"]).cuda()
descs = [("intensity", capi.DESC_INTENSITY, {}), ("laplacian", capi.DESC_LAPLACIAN, {}), ("gradient", capi.DESC_GRADIENT, {}), ("fields1", capi.DESC_FIELDS1, {}),
         ("fields2", capi.DESC_FIELDS2, {}), ("bitplanes", capi.DESC_BITPLANES, {}), ("latch 1 byte", capi.DESC_LATCH, dict(latchNumBytes=1)),
         ("latch 4 bytes", capi.DESC_LATCH, dict(latchNumBytes=4)), ("centraldiff r3", capi.DESC_CENTRAL_DIFFERENCE, dict(centralDifferenceRadius=3)),
         ("centraldiff r4 (80 ch: groups)", capi.DESC_CENTRAL_DIFFERENCE, dict(centralDifferenceRadius=4)), ("latch 16 bytes (128 ch: groups)", capi.DESC_LATCH, dict(latchNumBytes=16))]
print(f"{n} pairs {cols}x{rows}, {levels} levels, Huber; ns per point-channel-linearisation = step time / sum over pairs and levels of (points x channels x iterations)")
for name, d, kw in descs:
    for interp, iname in ((0, "kLinear"), (2, "kCubic")):
        p = hip.default_params(); p.numPyramidLevels = levels; p.descriptor = d; p.lossFunction = capi.LOSS_HUBER; p.verbosity = capi.VERB_SILENT; p.interp = interp
        ok = True
        for k, v in kw.items():
            if hasattr(p, k): setattr(p, k, v)
            else: ok = False
        if not ok:
            print(f"{name:34s} {iname}: parameter not in this build"); continue
        try:
            ctx = hip.create(batch["K"], batch["b"], rows, cols, p, device=0, n_frames=2 * n, n_pairs=n)
        except Exception as e:  # noqa: BLE001
            print(f"{name:34s} {iname}: {e}"); continue
        C = int(ctx.Cn)
        ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 3
        for _ in range(reps): poses, stats = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        its = np.asarray(stats["numIterations"], np.float64)[:, :levels]      # [pair][level]
        pts = np.array([[ctx.num_points(2 * i, l) for l in range(levels)] for i in range(n)], np.float64)
        work = float((its * pts).sum()) * max(1, C)
        print(f"{name:34s} {iname:7s}: C {C:3d}  {1e3 * dt:9.2f} ms per step, {its.sum() / dt / 1e3:8.1f} k GN it/s, points of pair 0 {pts[0].astype(int).tolist()}, {1e9 * dt / max(work, 1):7.3f} ns per point-channel-linearisation", flush=True)
        ctx.close()
