#!/usr/bin/env python3
"""What the validation mode costs: bpvo_hip_batch_run of n KITTI-shaped bit-planes pairs with option reference_reduction = 0 / 1, and a single pair.
python scripts/reference_mode_cost.py [pairs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpvo_amd
from bpvo_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = synth.make_batch(376, 1241, n, first_index=1000, workers=8)
import torch
torch.cuda.init(); dev = torch.device("cuda", 0)      # (torch's runtime first: tests/conftest.py)
hip = bpvo_amd.load()
for pairs in (n, 1):
    for mode in (0, 1):
        p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
        ctx = hip.create(b["K"], b["b"], 376, 1241, p, device=0, n_frames=2 * pairs, n_pairs=pairs)
        ctx.set_option("reference_reduction", mode)
        d_i, d_d = torch.from_numpy(b["images"][: 2 * pairs]).to(dev), torch.from_numpy(b["disparities"][: 2 * pairs]).to(dev)
        ctx.batch_run_device(pairs, d_i.data_ptr(), d_d.data_ptr())
        ctx.profiling(0); torch.cuda.synchronize()
        steps = 3
        t0 = time.perf_counter()
        for _ in range(steps): ctx.batch_run_device(pairs, d_i.data_ptr(), d_d.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        gn = ctx.total_linearizations() / steps
        print(f"{pairs} pairs, reference_reduction = {mode}: {1e3 * dt:.1f} ms per step, {gn / dt:.0f} GN it/s, {1e6 * dt * pairs / gn:.0f} us per linearisation per pair", flush=True)
        ctx.close()
