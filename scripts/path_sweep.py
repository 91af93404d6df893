#!/usr/bin/env python3
"""Does the library pick the faster path?  bpvo_hip_batch_run over a grid of workloads — descriptor (1 / 8 channels) x image size x NMS on / off x pairs per
call — with the default options (persistent kernel for one pair, team kernel for 2 - 128, chain otherwise, as the rules of estimate.hip decide) against the
four-kernel chain forced (team=0,persistent=0).  Prints ms per step of both and flags every cell where the default is more than 8 % slower.
    python scripts/path_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()      # (before the library loads: both initialise the HIP runtime)
from bpvo_amd import capi, synth
import bpvo_amd
hip = bpvo_amd.load()
flagged = []
for rows, cols, levels in ((240, 320, 3), (480, 640, 4), (376, 1241, 4)):
    batch = synth.make_batch(rows, cols, 128, first_index=0, workers=8)
    for desc, dn, loss in ((capi.DESC_INTENSITY, "intensity/huber", capi.LOSS_HUBER), (capi.DESC_BITPLANES, "bitplanes/tukey", capi.LOSS_TUKEY)):
        for nms in (1, 0):
            for n in (1, 2, 8, 32, 128):
                d_i, d_d = torch.from_numpy(batch["images"][: 2 * n]).cuda(), torch.from_numpy(batch["disparities"][: 2 * n]).cuda()
                ms, ref = {}, None
                for o in ("", "team=0,persistent=0"):
                    os.environ["BPVO_HIP_OPTIONS"] = o
                    p = hip.default_params(); p.numPyramidLevels = levels; p.descriptor = desc; p.lossFunction = loss; p.verbosity = capi.VERB_SILENT
                    if nms == 0: p.nonMaxSuppRadius = 0
                    ctx = hip.create(batch["K"], batch["b"], rows, cols, p, device=0, n_frames=2 * n, n_pairs=n)
                    poses, _ = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    reps = 3 if n >= 32 else 6
                    for _ in range(reps): poses, stats = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
                    torch.cuda.synchronize(); ms[o] = 1e3 * (time.perf_counter() - t0) / reps
                    if ref is None: ref = poses.copy()
                    same = np.array_equal(ref.view(np.uint32), poses.view(np.uint32))
                    pts = [max(ctx.num_points(2 * i, l) for i in range(n)) for l in range(levels)]
                    ctx.close()
                slow = ms[""] > 1.08 * ms["team=0,persistent=0"]
                line = f"{cols}x{rows} {dn} NMS {'on' if nms else 'off'} {n:3d} pairs: default {ms['']:8.3f} ms, chain forced {ms['team=0,persistent=0']:8.3f} ms, ratio {ms[''] / ms['team=0,persistent=0']:.2f}, same bits {same}, most points per level {pts}" + ("   <-- SLOWER" if slow else "")
                print(line, flush=True)
                if slow: flagged.append(line)
print("cells where the default path is more than 8 % slower than the forced chain:", len(flagged))
