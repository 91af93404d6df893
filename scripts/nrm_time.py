import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bpvo_amd
from bpvo_amd import capi, synth
ROOT=os.getcwd()
hip = capi.Binding(os.path.join(ROOT, os.environ["BPVO_AB_LIB"]), "bpvo_hip_") if os.environ.get("BPVO_AB_LIB") else bpvo_amd.load()
rows, cols = 376, 1241
b = synth.make_batch(rows, cols, 1, first_index=0, workers=1)
p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2, n_pairs=1)
ctx.frame_set_data(0, b["images"][0], b["disparities"][0])
for rep in range(3):
    t0=time.perf_counter()
    for k in range(20): ctx.frame_set_template(0)
    dt=(time.perf_counter()-t0)/20
print(os.environ.get("BPVO_AB_LIB","default"), "set_template %.1f us" % (1e6*dt))
