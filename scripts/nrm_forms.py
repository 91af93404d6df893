#!/usr/bin/env python3
"""The four forms of the sequential Hartley sums (option "normalization_form"): time of the normalisation launch by HIP events (kernel class
"normalization", profiling level 2) for a 1241x376 template with NMS (26 k points at most per level) and a dense 640x480 one (300 k), and the
bits of (s, c) of every level compared.   python scripts/nrm_forms.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpvo_amd
from bpvo_amd import capi, synth
hip = bpvo_amd.load()
for name, rows, cols, levels, kw in (("1241x376 bit-planes, NMS (sparse template)", 376, 1241, 4, {}),
                                     ("640x480 intensity, NMS off (dense template)", 480, 640, 3, dict(nonMaxSuppRadius=0, minSaliency=0.001, descriptor=capi.DESC_INTENSITY))):
    b = synth.make_batch(rows, cols, 1, first_index=0, workers=1)
    ref = None
    for form in (2, 1, 0, 3):
        p = hip.default_params(); p.numPyramidLevels = levels; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
        for k, v in kw.items(): setattr(p, k, v)
        ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2, n_pairs=1)
        ctx.set_option("normalization_form", form)
        ctx.set_option("normalization_side_stream", 0)
        ctx.frame_set_data(0, b["images"][0], b["disparities"][0])
        for _ in range(3): ctx.frame_set_template(0)
        ctx.profiling(2)
        for _ in range(10): ctx.frame_set_template(0)
        ks = {k["name"]: k for k in ctx.kernel_stats()}["normalization"]
        nrm = [np.stack(ctx.get_normalization(0, l)) for l in range(levels)]
        if ref is None: ref = nrm
        same = all(np.array_equal(a.view(np.uint32), r.view(np.uint32)) for a, r in zip(nrm, ref))
        print(f"{name}: form {form}: {1e3 * ks['total_ms'] / max(1, ks['launches']):.1f} us per launch ({int(ks['launches'])} launches), points per level {[ctx.num_points(0, l) for l in range(levels)]}, bits equal to form 2: {same}", flush=True)
        ctx.close()
