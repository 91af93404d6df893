#!/usr/bin/env python3
"""Single pair, dense templates (NMS off): estimate_pose ms with the persistent kernel against the four-kernel chain as the template grows —
where option "persist_max_points" should sit.   python scripts/persist_crossover.py"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpvo_amd
from bpvo_amd import capi, synth
hip = bpvo_amd.load()
for desc, dn in ((capi.DESC_BITPLANES, "bitplanes"), (capi.DESC_INTENSITY, "intensity")):
    for rows, cols in ((120, 160), (180, 240), (240, 320), (300, 400), (360, 480), (480, 640), (376, 1241)):
        d = synth.make_pair(rows, cols, 3)
        res = {}
        for mode in ("persistent", "chain"):
            p = hip.default_params(); p.numPyramidLevels = 1; p.descriptor = desc; p.lossFunction = capi.LOSS_HUBER; p.verbosity = capi.VERB_SILENT
            p.nonMaxSuppRadius = 0; p.minSaliency = 0.001; p.minNumPixelsForNonMaximaSuppression = 10 ** 9
            ctx = hip.create(d["K"], d["b"], rows, cols, p, device=0, n_frames=2, n_pairs=1)
            ctx.set_option("persistent", 1 if mode == "persistent" else 0)
            ctx.frame_set_data(0, d["imgA"], d["dispA"]); ctx.frame_set_template(0); ctx.frame_set_data(1, d["imgB"], d["dispB"])
            for _ in range(3): T, st = ctx.estimate_pose(0, 0, 1)
            t0 = time.perf_counter()
            for _ in range(10): T, st = ctx.estimate_pose(0, 0, 1)
            dt = (time.perf_counter() - t0) / 10
            res[mode] = (1e3 * dt, st[0]["numIterations"], ctx.num_points(0, 0), T.copy())
            ctx.close()
        assert np.array_equal(res["persistent"][3].view(np.uint32), res["chain"][3].view(np.uint32))
        n = res["chain"][2]; it = res["chain"][1] + 2
        print(f"{dn:10s} {cols}x{rows}: {n:7d} points, {it} linearisations | persistent {res['persistent'][0]:.3f} ms ({1e3 * res['persistent'][0] / it:.1f} us/lin) | chain {res['chain'][0]:.3f} ms ({1e3 * res['chain'][0] / it:.1f} us/lin)", flush=True)
