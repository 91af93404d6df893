"""Latency of the single-pair path (one 1241x376 bit-planes pair per call, the B = 1 case) and of sequential addFrame."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpvo_amd
from bpvo_amd import capi, synth

hip = bpvo_amd.load()
rows, cols = 376, 1241
b = synth.make_batch(rows, cols, 4, first_index=0, workers=1)
p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2, n_pairs=1)
for rep in range(3):
    t0 = time.perf_counter(); its = 0
    for k in range(4):
        poses, stats = ctx.batch_run(b["images"][2 * k: 2 * k + 2], b["disparities"][2 * k: 2 * k + 2])
        its += int(stats["numIterations"].sum())
    dt = time.perf_counter() - t0
print("batch_run B=1: %.2f ms per pair (host buffers), %d GN iterations per pair" % (1e3 * dt / 4, its // 4))
ctx.frame_set_data(0, b["images"][0], b["disparities"][0]); ctx.frame_set_template(0); ctx.frame_set_data(1, b["images"][1], b["disparities"][1])
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(10):
        T, st = ctx.estimate_pose(0, 0, 1)
    dt = time.perf_counter() - t0
n_it = sum(s["numIterations"] for s in st)
print("estimate_pose only: %.2f ms per call, %d iterations -> %.1f us per GN iteration" % (1e3 * dt / 10, n_it, 1e6 * dt / 10 / max(1, n_it)))
# sequential VO
seq = synth.make_sequence(rows, cols, 12, index=3)
frames = seq["frames"]
vo = hip.create(seq["K"], seq["b"], rows, cols, p, n_frames=3, n_pairs=1)
t0 = time.perf_counter()
for img, disp in frames:
    vo.add_frame(img, disp)
dt = time.perf_counter() - t0
print("addFrame: %.2f ms per frame over %d frames" % (1e3 * dt / len(frames), len(frames)))
