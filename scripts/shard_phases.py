#!/usr/bin/env python3
"""Where a batch's step goes when its stages are run ONE AFTER THE OTHER on one lane, each followed by a synchronisation:
setData (pyramid + descriptors), setTemplate (saliency, selection, normalisation, template build), estimate (the Gauss-Newton stage),
next to bpvo_hip_batch_run (staggered lanes).  BPVO_HIP_OPTIONS=lanes=1 python scripts/shard_phases.py --pairs 128"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--rows", type=int, default=376)
    ap.add_argument("--cols", type=int, default=1241)
    a = ap.parse_args()
    import numpy as np
    import torch
    import bpvo_amd
    from bpvo_amd import capi, synth
    n = a.pairs
    path = f"/tmp/shard_ab_{a.rows}x{a.cols}_{n}.npz"
    if os.path.exists(path):
        d = np.load(path)
        b = dict(images=d["images"], disparities=d["disparities"], K=d["K"], b=float(d["b"]))
    else:
        b = synth.make_batch(a.rows, a.cols, n, first_index=0, workers=min(16, os.cpu_count() or 1))
        np.savez(path, images=b["images"], disparities=b["disparities"], K=b["K"], b=b["b"])
    hip = bpvo_amd.load()
    p = hip.default_params()
    p.numPyramidLevels = 4
    p.descriptor = capi.DESC_BITPLANES
    p.lossFunction = capi.LOSS_TUKEY
    p.verbosity = capi.VERB_SILENT
    ctx = hip.create(b["K"], float(b["b"]), a.rows, a.cols, p, device=0, n_frames=2 * n, n_pairs=n)
    dev = torch.device("cuda", 0)
    di = torch.from_numpy(b["images"][: 2 * n]).to(dev)
    dd = torch.from_numpy(b["disparities"][: 2 * n]).to(dev)
    sync = torch.cuda.synchronize
    for _ in range(2):
        ctx.batch_run_device(n, di.data_ptr(), dd.data_ptr())
    acc = dict(set_data=0.0, set_template=0.0, estimate=0.0, batch_run=0.0)
    for _ in range(a.steps):
        sync(); t0 = time.perf_counter()
        ctx.frames_set_data_device(0, 1, 2 * n, di.data_ptr(), dd.data_ptr())
        sync(); t1 = time.perf_counter()
        ctx.frames_set_template(0, 2, n)
        sync(); t2 = time.perf_counter()
        ctx.batch_estimate(n)
        sync(); t3 = time.perf_counter()
        ctx.batch_run_device(n, di.data_ptr(), dd.data_ptr())
        sync(); t4 = time.perf_counter()
        acc["set_data"] += t1 - t0; acc["set_template"] += t2 - t1; acc["estimate"] += t3 - t2; acc["batch_run"] += t4 - t3
    poses, stats = ctx.batch_run_device(n, di.data_ptr(), dd.data_ptr())
    its = stats["numIterations"]          # [pairs][levels]
    for l in range(its.shape[1] - 1, -1, -1):
        v = np.sort(its[:, l])
        print(f"level {l}: iterations min {v[0]} p25 {v[len(v) // 4]} median {v[len(v) // 2]} p75 {v[3 * len(v) // 4]} p90 {v[9 * len(v) // 10]} max {v[-1]} mean {v.mean():.1f}")
    ms = {k: 1e3 * v / a.steps for k, v in acc.items()}
    print(f"{n} pairs, options {os.environ.get('BPVO_HIP_OPTIONS', 'default')}: set_data {ms['set_data']:.2f} ms, set_template {ms['set_template']:.2f} ms, "
          f"estimate {ms['estimate']:.2f} ms, sum {ms['set_data'] + ms['set_template'] + ms['estimate']:.2f} ms | batch_run {ms['batch_run']:.2f} ms")


if __name__ == "__main__":
    main()
