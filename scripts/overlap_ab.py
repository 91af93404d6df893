#!/usr/bin/env python3
"""(Needs profiles/r06_team_overlap_rejected.diff applied: the option does not exist in the library as shipped.)  A/B of option "team_overlap_finest_level" on the config-5 shards (pairs of the 1024-pair batch on one GPU): GN iterations/s and ms per step of
bpvo_hip_batch_run with the option off / on, poses and statistics compared bit for bit.   python scripts/overlap_ab.py [pairs ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [128, 96, 64, 48]
    from bpvo_amd import capi, synth
    import bpvo_amd
    nmax = max(sizes)
    batch = synth.make_batch(376, 1241, nmax, first_index=1000, workers=8)
    import torch
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    hip = bpvo_amd.load()
    for n in sizes:
        res = {}
        for opt in (0, 1, 0, 1):
            p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
            ctx = hip.create(batch["K"], batch["b"], 376, 1241, p, device=0, n_frames=2 * n, n_pairs=n)
            ctx.set_option("team_overlap_finest_level", opt)
            d_i, d_d = torch.from_numpy(batch["images"][: 2 * n]).to(dev), torch.from_numpy(batch["disparities"][: 2 * n]).to(dev)
            for _ in range(3):
                ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
            ctx.profiling(0)
            torch.cuda.synchronize()
            steps = 12
            t0 = time.perf_counter()
            for _ in range(steps):
                poses, stats = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            gn = ctx.total_linearizations()
            lv, gave_up = ctx.persistent_counts()
            res.setdefault(opt, []).append((gn / dt, 1e3 * dt / steps, poses.copy(), stats.copy(), ctx.team_counts(), gave_up))
            ctx.close()
        same = all(np.array_equal(res[0][0][2].view(np.uint32), r[2].view(np.uint32)) and np.array_equal(res[0][0][3], r[3]) for k in res for r in res[k])
        print(json.dumps({"pairs": n, "off_gn_it_per_s": [round(r[0]) for r in res[0]], "on_gn_it_per_s": [round(r[0]) for r in res[1]],
                          "off_ms": [round(r[1], 2) for r in res[0]], "on_ms": [round(r[1], 2) for r in res[1]],
                          "gain": round(np.mean([r[0] for r in res[1]]) / np.mean([r[0] for r in res[0]]), 4),
                          "team_launches_on": res[1][0][4], "gave_up": [r[5] for k in res for r in res[k]], "bit_identical": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
